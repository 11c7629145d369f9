"""CPU: host-side mirror of the reference interface (no GPU compute): padding / normalisation / naming helpers against
reference-generated vectors, the checkpoint key/shape contract, TIFF I/O, and that libmseg_hip.so loads and exports
every symbol declared in include/mseg_hip.h."""
import ctypes
import json
import pathlib
import re

import numpy as np
import pytest
import torch

from helpers import GOLDEN, load_npz

ROOT = pathlib.Path(__file__).resolve().parents[1]


def test_zero_pad_and_frame_normalisation_match_reference():
    from microbeseg_amd.utils.utils import zero_pad_model_input, min_max_normalization
    fx = load_npz("host_contract.npz")
    for name, shape in (("200x300", (200, 300)), ("65x64", (65, 64)), ("2048x2048", (2048, 2048)), ("321x1000", (321, 1000))):
        img = np.zeros(shape, np.uint16)
        padded, pads = zero_pad_model_input(img, pad_val=3)
        assert list(padded.shape) == list(fx[f"pad_{name}_shape"]) and pads == list(fx[f"pad_{name}_pads"])
        assert (padded[:pads[0]] == 3).all() and (padded[:, :pads[1]] == 3).all()      # TOP / LEFT padding
    img = fx["pad_65x64_in"]
    padded, pads = zero_pad_model_input(img, pad_val=img.min())
    assert np.array_equal(padded, fx["pad_65x64_out"])
    fmin, fmax = np.min(img), np.max(img)
    norm = (2 * (padded.astype(np.float32) - fmin) / (fmax - fmin) - 1).astype(np.float32)
    assert np.array_equal(norm, fx["norm_65x64"])
    assert np.array_equal(min_max_normalization(fx["mmn_in"], min_value=0, max_value=65535), fx["mmn_out"])
    with pytest.raises(Exception):
        zero_pad_model_input(np.zeros((9000, 9000), np.uint8))


def test_state_dict_contract_matches_reference():
    """Checkpoints are exchanged with the reference: identical keys, order, shapes and dtypes."""
    from microbeseg_amd.utils.unets import build_unet
    with open(GOLDEN / "state_dict_keys.json") as f:
        want = json.load(f)
    for tag, keys in want.items():
        ut, norm, f0, f1 = tag.split("_")
        net = build_unet(ut, "relu", "conv", norm, "cpu", 1, ch_out=3 if ut == "U" else 1, filters=(int(f0), int(f1)))
        got = {k: [list(v.shape), str(v.dtype)] for k, v in net.state_dict().items()}
        assert list(got) == list(keys), tag
        assert got == keys, tag
    with pytest.raises(Exception):
        build_unet("X", "relu", "conv", "bn", "cpu", 1)
    with pytest.raises(Exception):
        build_unet("U", "tanh", "conv", "bn", "cpu", 1)


def test_cpu_tensor_is_rejected_loudly():
    """No silent CPU fallback in the product path."""
    from microbeseg_amd.utils.unets import build_unet
    net = build_unet("DU", "relu", "conv", "bn", "cpu", 1, filters=(8, 16))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.zeros(1, 1, 32, 32))


def test_unique_path_max_epochs_and_train_info(tmp_path):
    from microbeseg_amd.utils.utils import unique_path, write_train_info
    from microbeseg_amd.training.train import get_max_epochs
    assert unique_path(tmp_path, "distance_model_{:02d}.pth").name == "distance_model_01.pth"
    (tmp_path / "distance_model_01.pth").write_bytes(b"")
    assert unique_path(tmp_path, "distance_model_{:02d}.pth").name == "distance_model_02.pth"
    # reference heuristic (train.py:579-606): table x sqrt(320 / crop) floored to a multiple of 20
    assert [get_max_epochs(n, 320) for n in (1000, 500, 200, 100, 50, 49)] == [200, 240, 320, 400, 480, 560]
    assert get_max_epochs(40, 256) == 620 and get_max_epochs(1200, 256) == 220
    write_train_info({"run_name": "m", "architecture": ("DU", "conv", "relu", "bn", [8, 16])}, tmp_path)
    assert json.load(open(tmp_path / "m.json"))["architecture"][4] == [8, 16]


def test_tiff_roundtrip(tmp_path):
    from microbeseg_amd.utils import tiffio
    rng = np.random.default_rng(0)
    for arr in (rng.integers(0, 65535, (33, 47)).astype(np.uint16), rng.random((20, 31)).astype(np.float32),
                rng.integers(0, 3, (16, 16)).astype(np.uint8), rng.integers(0, 900, (3, 24, 40)).astype(np.uint16)):
        p = tmp_path / "x.tif"
        tiffio.imwrite(p, arr)
        back = tiffio.imread(p)
        assert back.dtype == arr.dtype and np.array_equal(back, arr)


def test_library_exports_every_declared_symbol():
    """The C-ABI shared object loads (no compute calls) and exports what include/mseg_hip.h declares."""
    from microbeseg_amd import _lib
    header = (ROOT / "include" / "mseg_hip.h").read_text()
    declared = set(re.findall(r"\b(mseg_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.mseg_version() >= 100
    assert lib.mseg_strerror(-1).decode().startswith("invalid")
    assert lib.mseg_norm_workspace_bytes(2, 1024, 64) > 0 and lib.mseg_postproc_workspace_bytes(64, 64) > 0
    assert lib.mseg_igemm(None, None) == -1          # argument validation happens before any GPU work


def test_augmentation_decision_tree_frequencies():
    """draw_parameters follows the reference's probabilities (mytransforms.py:24-29): Flip 1.0 over 8 symmetries,
    Contrast 0.45, Scaling 0.25, Rotate 0.25, Blur 0.3, Noise 0.3, and the parameter ranges."""
    import random
    from microbeseg_amd.training.device_augment import draw_parameters, scale_matrices, rotation_matrices
    p = draw_parameters(20000, random.Random(3), np.random.default_rng(3))
    assert set(np.unique(p["flip"])) == set(range(8))
    for key, want in (("scale_apply", 0.25), ("rot_apply", 0.25)):
        assert abs(p[key].mean() - want) < 0.015
    assert abs((p["contrast"][:, 0] > 0).mean() - 0.45) < 0.015
    assert abs((p["contrast"][:, 0] == 2).mean() - 0.15) < 0.01 and abs((p["contrast"][:, 0] == 3).mean() - 0.15) < 0.01
    assert abs((p["blur_sigma"] > 0).mean() - 0.3) < 0.015 and abs((p["noise_frac"] > 0).mean() - 0.3) < 0.015
    b = p["blur_sigma"][p["blur_sigma"] > 0]
    assert b.min() >= 1.0 and b.max() < 2.0
    assert set(np.round(p["noise_frac"][p["noise_frac"] > 0] * 100).astype(int)) == {1, 2, 3, 4, 5}
    s = p["scale_xy"][p["scale_apply"] > 0]
    assert s.min() >= 0.85 and s.max() <= 1.15 and abs(p["rot_deg"]).max() <= 45
    g = p["contrast"][p["contrast"][:, 0] == 2]
    assert g[:, 1].min() >= 0.75 and g[:, 1].max() <= 1.25 and g[:, 2].min() >= 0.7 and g[:, 2].max() <= 1.3
    # the centre is a fixed point of both warps
    m = np.concatenate([scale_matrices(np.array([[1.1, 0.9]]), 31, 41), rotation_matrices(np.array([33.0]), 31, 41)])
    for q in m:
        assert abs(q[0] * 20 + q[1] * 15 + q[2] - 20) < 1e-4 and abs(q[3] * 20 + q[4] * 15 + q[5] - 15) < 1e-4

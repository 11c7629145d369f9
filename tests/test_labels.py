"""Label creation: oracle vs vectors from the real reference (CPU), HIP kernels vs the same vectors
(GPU; exact for the boundary / border labels, 1e-6 absolute for the distance labels — integer distances and decisions are
exact, the fp64 rescaling goes through the device's exp)."""
import pathlib
import sys

import numpy as np
import pytest
import torch

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from oracle import labels_ref  # noqa: E402

G = np.load(ROOT / "tests" / "golden" / "labels_boundary.npz")
CASES = sorted(k[1:] for k in G.files if k.startswith("m"))


@pytest.mark.parametrize("i", CASES)
def test_oracle_matches_reference(i):
    assert np.array_equal(labels_ref.boundary_label(G[f"m{i}"]), G[f"boundary{i}"])
    assert np.array_equal(labels_ref.border_label(G[f"m{i}"]), G[f"border{i}"])


@pytest.mark.gpu
@pytest.mark.parametrize("i", CASES)
def test_hip_matches_reference(i):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from microbeseg_amd.training import train_data_representations as T
    b = T.get_label(G[f"m{i}"], "boundary", 0)
    assert b.dtype == np.uint8 and np.array_equal(b, G[f"boundary{i}"])
    assert np.array_equal(T.border_label(G[f"m{i}"]), G[f"border{i}"])


@pytest.mark.gpu
def test_hip_matches_oracle_random():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from microbeseg_amd.training import train_data_representations as T
    rng = np.random.default_rng(12)
    m = np.kron(rng.integers(0, 5, (40, 50)), np.ones((5, 7), int)).astype(np.uint16)
    m[rng.random(m.shape) < 0.05] = 0
    assert np.array_equal(T.boundary_label(m), labels_ref.boundary_label(m))
    assert np.array_equal(T.border_label(m), labels_ref.border_label(m))
    with pytest.raises(RuntimeError):
        T.get_label(m, "adapted_border", 20)


# ---- distance labels ---------------------------------------------------------------------------------------------------
GD = np.load(ROOT / "tests" / "golden" / "labels_distance.npz")
DCASES = sorted(k[1:] for k in GD.files if k.startswith("m") and k[1:].isdigit())
DTOL = 1e-6


def _cells(rng, H, W, n, rmin, rmax, gap):
    """random ellipses, optionally shrunk so that narrow background gaps separate neighbours"""
    from scipy import ndimage as ndi
    m = np.zeros((H, W), np.uint16)
    yy, xx = np.mgrid[0:H, 0:W]
    for k in range(1, n + 1):
        cy, cx = rng.uniform(0, H), rng.uniform(0, W)
        a, b, th = rng.uniform(rmin, rmax), rng.uniform(rmin, rmax), rng.uniform(0, np.pi)
        u = (yy - cy) * np.cos(th) + (xx - cx) * np.sin(th)
        v = -(yy - cy) * np.sin(th) + (xx - cx) * np.cos(th)
        m[((u / a) ** 2 + (v / b) ** 2 <= 1) & (m == 0)] = k
    if gap:
        out = np.zeros_like(m)
        for k in np.unique(m)[1:]:
            out[ndi.binary_erosion(m == k, iterations=gap)] = k
        m = out
    return m


@pytest.mark.parametrize("i", DCASES)
def test_distance_oracle_matches_reference(i):
    comp, weight = labels_ref.bottom_hat_closing(GD[f"m{i}"])
    assert np.array_equal(comp > 0, GD[f"closed{i}"] > 0)
    assert np.array_equal(weight, GD[f"corr{i}"])
    cell, nb = labels_ref.distance_label(GD[f"m{i}"], int(GD[f"sr{i}"]))
    assert np.array_equal(cell, GD[f"cell{i}"])
    assert np.array_equal(nb, GD[f"neighbor{i}"])


def test_distance_golden_covers_branches():
    rim = sum(int((GD[f"corr{i}"] == np.float32(0.8)).sum()) for i in DCASES)
    touching = sum(int((labels_ref.border_label(GD[f"m{i}"]) == 2).sum()) for i in DCASES)
    assert rim > 100 and touching > 100


@pytest.mark.gpu
@pytest.mark.parametrize("i", DCASES)
def test_bottom_hat_closing_hip_matches_reference(i):
    """bottom_hat_closing (reference :40-72) as its own entry point: the reference's closed / corr images of the golden masks
    (committed fixtures) and the oracle's component numbering (scipy / measure.label order), bit for bit."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from microbeseg_amd.training import train_data_representations as T
    closed, corr = T.bottom_hat_closing(GD[f"m{i}"])
    assert corr.dtype == np.float32
    assert np.array_equal(closed, GD[f"closed{i}"])            # the reference's measure.label ids
    assert np.array_equal(corr, GD[f"corr{i}"])
    comp, weight = labels_ref.bottom_hat_closing(GD[f"m{i}"])
    assert np.array_equal(closed, comp) and np.array_equal(corr, weight)


@pytest.mark.gpu
@pytest.mark.parametrize("i", DCASES)
def test_distance_hip_matches_reference(i):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from microbeseg_amd.training import train_data_representations as T
    cell, nb = T.distance_label(GD[f"m{i}"], int(GD[f"sr{i}"]))
    assert cell.dtype == np.float32 and nb.dtype == np.float32
    assert np.abs(cell - GD[f"cell{i}"]).max() <= DTOL
    assert np.abs(nb - GD[f"neighbor{i}"]).max() <= DTOL


@pytest.mark.gpu
def test_distance_hip_matches_oracle_random_batch():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from microbeseg_amd.training import train_data_representations as T
    rng = np.random.default_rng(77)
    masks = np.stack([_cells(rng, 96, 112, n, 4, 11, gap) for n, gap in ((25, 1), (12, 0), (40, 1), (6, 2), (0, 0))])
    masks[1][masks[1] == 3] = 40000                       # ids are arbitrary uint16 values
    masks[3][:, :4][masks[3][:, :4] > 0] = 7              # a cell in pieces along the frame
    for sr in (9, 30):
        cell, nb = T.distance_label_batch(masks, sr)
        for j in range(len(masks)):
            c, d = labels_ref.distance_label(masks[j], sr)
            assert np.abs(cell[j] - c).max() <= DTOL, (sr, j)
            assert np.abs(nb[j] - d).max() <= DTOL, (sr, j)
    # get_label dispatch: search radius = ceil(0.75 * max_mal)
    c, d = T.get_label(masks[0], "distance", 40)
    c2, d2 = labels_ref.distance_label(masks[0], 30)
    assert np.abs(c - c2).max() <= DTOL and np.abs(d - d2).max() <= DTOL
    with pytest.raises(RuntimeError):
        T.distance_label(masks, 10)


@pytest.mark.parametrize("i", DCASES)
def test_major_axis_oracle_matches_reference(i):
    mal = labels_ref.major_axis_lengths(GD[f"m{i}"])
    ref = GD[f"mal{i}"]
    assert mal.shape == ref.shape and np.allclose(mal, ref, rtol=1e-12, atol=0)
    assert int(np.ceil(mal.max())) == int(np.ceil(ref.max()))


@pytest.mark.gpu
def test_major_axis_hip_matches_reference():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from microbeseg_amd.training import train_data_representations as T
    for i in DCASES:
        v = T.max_major_axis_length(GD[f"m{i}"])
        assert abs(v - GD[f"mal{i}"].max()) <= 1e-9 * GD[f"mal{i}"].max()
    assert T.max_major_axis_length(np.zeros((16, 16), np.uint16)) == 0.0


@pytest.mark.gpu
def test_create_labels_worker(tmp_path):
    """CreateLabelsWorker (reference train.py:26-112): files, dtypes and values of both methods; stop handling."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from microbeseg_amd.utils import synth, tiffio
    from src.training.train import CreateLabelsWorker
    root = synth.write_training_set(tmp_path / "set", 3, 2, size=96, seed=9, label_types=())
    w = CreateLabelsWorker()
    prog, msgs, done = [], [], []
    w.progress.connect(prog.append)
    w.text_output.connect(msgs.append)
    w.finished.connect(lambda: done.append(1))
    w.create_labels(root, "distance")
    assert done == [1] and prog[-1] == 100 and msgs[0] == "Create labels"
    for mode, n in (("train", 3), ("val", 2)):
        for i in range(n):
            mask = tiffio.imread(root / mode / f"mask_{i:03d}.tif")
            sr = int(np.ceil(0.75 * int(np.ceil(labels_ref.major_axis_lengths(mask).max()))))
            c, d = labels_ref.distance_label(mask, sr)
            cell = tiffio.imread(root / mode / f"cell_dist_{i:03d}.tif")
            nb = tiffio.imread(root / mode / f"neighbor_dist_{i:03d}.tif")
            assert cell.dtype == np.float32 and nb.dtype == np.float32
            assert np.abs(cell - c).max() <= DTOL and np.abs(nb - d).max() <= DTOL
    w.create_labels(root, "boundary")
    b = tiffio.imread(root / "val" / "boundary_001.tif")
    assert b.dtype == np.uint8 and np.array_equal(b, labels_ref.boundary_label(tiffio.imread(root / "val" / "mask_001.tif")))
    # fewer than two validation masks: message, nothing done
    (root / "val" / "mask_001.tif").unlink()
    prog.clear(), msgs.clear()
    w.create_labels(root, "distance")
    assert prog == [0] and "at least two annotated" in msgs[-1]
    # stop request: the folder is deleted, like the reference does
    root2 = synth.write_training_set(tmp_path / "set2", 2, 2, size=64, seed=3, label_types=())
    w2 = CreateLabelsWorker()
    w2.stop_label_creation_process()
    w2.create_labels(root2, "distance")
    assert not root2.exists()


@pytest.mark.parametrize("i", DCASES)
def test_cell_distance_oracle_matches_reference(i):
    m, sr = GD[f"m{i}"], int(GD[f"sr{i}"])
    assert np.array_equal(labels_ref.cell_distance_label(m, sr), GD[f"celld{i}"])
    assert np.array_equal(labels_ref.cell_distance_label(m, sr, apply_clipping=True), GD[f"cellc{i}"])


@pytest.mark.gpu
@pytest.mark.parametrize("i", DCASES)
def test_cell_distance_hip_matches_reference(i):
    """label types 'cell_dist' / 'cell_dist_clipped' (cell_distance_label, reference :219-258)"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from microbeseg_amd.training import train_data_representations as T
    m, sr = GD[f"m{i}"], int(GD[f"sr{i}"])
    assert np.abs(T.cell_distance_label(m, sr) - GD[f"celld{i}"]).max() <= DTOL
    assert np.abs(T.cell_distance_label(m, sr, apply_clipping=True) - GD[f"cellc{i}"]).max() <= DTOL
    # dispatcher: the search radius is ceil(0.75 * max_mal), so max_mal = sr / 0.75 rounded down reproduces sr
    mm = int(sr / 0.75)
    if int(np.ceil(0.75 * mm)) == sr:
        assert np.abs(T.get_label(m, "cell_dist", mm) - GD[f"celld{i}"]).max() <= DTOL
        assert np.abs(T.get_label(m, "cell_dist_clipped", mm) - GD[f"cellc{i}"]).max() <= DTOL


@pytest.mark.parametrize("i", DCASES)
def test_j4_oracle_matches_reference(i):
    assert np.array_equal(labels_ref.j4_label(GD[f"m{i}"]), GD[f"j4{i}"])


@pytest.mark.gpu
@pytest.mark.parametrize("i", DCASES)
def test_j4_hip_matches_reference(i):
    """label type 'j4' (Pena et al.): background / cell / touching / gap, exact"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from microbeseg_amd.training import train_data_representations as T
    got = T.get_label(GD[f"m{i}"], "j4", 0)
    assert got.dtype == np.uint8 and np.array_equal(got, GD[f"j4{i}"])
    m = GD[f"m{i}"]
    assert np.array_equal(T.j4_label(m, k_neighbors=1, se_radius=2), labels_ref.j4_label(m, 1, 2))
    with pytest.raises(RuntimeError):
        T.get_label(m, "adapted_border", 0)

"""GPU (MI355X): the two local-file CLIs end to end as subprocesses — train_script.py (reference flags + --train_path)
then infer_script_local.py on a small 2D+t stack with the checkpoint it wrote."""
import pathlib
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = pathlib.Path(__file__).resolve().parents[1]


def test_train_then_infer_cli(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from microbeseg_amd.utils import synth, tiffio
    data = synth.write_training_set(tmp_path / "set", 24, 6, size=128, seed=5, label_types=("distance",))
    r = subprocess.run([sys.executable, str(ROOT / "train_script.py"), "--train_path", str(data), "-b", "8", "-i", "1",
                        "-m", "distance", "-o", "adam", "-r", str(tmp_path / "models"), "--filters", "8", "16",
                        "--max_epochs", "4"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    model = tmp_path / "models" / "set" / "distance_model_01"
    assert model.with_suffix(".pth").is_file() and model.with_suffix(".json").is_file()
    rng = np.random.Generator(np.random.PCG64(9))
    stack = np.stack([synth.synth_crop(rng, 128)["img"] for _ in range(5)])      # 2D+t: (T, H, W)
    imgs = tmp_path / "imgs"
    imgs.mkdir()
    tiffio.imwrite(str(imgs / "movie.tif"), stack)
    res = tmp_path / "results"
    r = subprocess.run([sys.executable, str(ROOT / "infer_script_local.py"), "-i", str(imgs), "-m", str(model), "-r",
                        str(res), "--rois"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = tiffio.imread(str(res / "mask_movie_channel0.tif"))
    assert out.shape == (5, 128, 128) and out.dtype == np.uint16
    # --rois: one polygon record per instance and frame (the OMERO route's payload, infer.py:265-287)
    import json
    rois = json.load(open(res / "mask_movie_channel0_rois.json"))["rois"]
    assert len(rois) == sum(len(np.unique(f)) - 1 for f in out)
    for roi in rois[:20]:
        pts = [tuple(int(v) for v in p.split(",")) for p in roi["points"].split()]
        frame = out[roi["theT"]]
        ids = {int(frame[y, x]) for x, y in pts}
        assert len(ids) == 1 and 0 not in ids and roi["theC"] == 0 and roi["strokeColor"] == -65281

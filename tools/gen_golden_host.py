#!/usr/bin/env python3
"""Generate tests/golden/host_contract.npz + state_dict_keys.json from the REAL reference (build container only):
padding / normalisation (src/utils/utils.py), epoch heuristic constants restated from train.py:579-606 are checked
against values computed by the reference formula, and the state-dict key/shape contract (src/utils/unets.py)."""
import json
import pathlib
import sys

import numpy as np

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
from src.utils.utils import zero_pad_model_input, min_max_normalization  # noqa: E402
from src.utils.unets import build_unet  # noqa: E402

OUT = pathlib.Path(__file__).resolve().parents[1] / "tests" / "golden"
rng = np.random.Generator(np.random.PCG64(77))
out = {}
for name, shape in (("200x300", (200, 300)), ("65x64", (65, 64)), ("2048x2048", (2048, 2048)), ("321x1000", (321, 1000))):
    img = rng.integers(100, 4000, shape).astype(np.uint16)
    padded, pads = zero_pad_model_input(img, pad_val=img.min())
    out[f"pad_{name}_shape"] = np.array(padded.shape)
    out[f"pad_{name}_pads"] = np.array(pads)
    if shape[0] * shape[1] < 5000:
        out[f"pad_{name}_in"] = img
        out[f"pad_{name}_out"] = padded
        fmin, fmax = np.min(img), np.max(img)
        # frame normalisation exactly as infer.py:346-348 (numpy scalar min / max of the image dtype)
        out[f"norm_{name}"] = (2 * (padded.astype(np.float32) - fmin) / (fmax - fmin) - 1).astype(np.float32)
crop = rng.integers(0, 65536, (32, 32, 1)).astype(np.uint16)
out["mmn_in"] = crop
out["mmn_out"] = min_max_normalization(crop, min_value=0, max_value=65535)
np.savez_compressed(OUT / "host_contract.npz", **out)

keys = {}
for ut, norm, filters, ch_out in (("DU", "bn", (64, 1024), 1), ("U", "bn", (64, 1024), 3), ("DU", "gn", (32, 512), 1),
                                  ("U", "in", (32, 256), 3), ("DU", "bn", (8, 16), 1)):
    net = build_unet(ut, "relu", "conv", norm, "cpu", 1, ch_out=ch_out, filters=filters)
    keys[f"{ut}_{norm}_{filters[0]}_{filters[1]}"] = {k: [list(v.shape), str(v.dtype)] for k, v in net.state_dict().items()}
with open(OUT / "state_dict_keys.json", "w") as f:
    json.dump(keys, f)
print("ok", {k: len(v) for k, v in keys.items()})

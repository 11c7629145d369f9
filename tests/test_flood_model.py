"""The closed form of the reference heap's marker phase on a constant image (tools/flood_stream_model.py) against the textbook
heap, pop for pop and push for push — the derivation csrc/postproc.hip::pp_flood_const_stream_kernel implements."""
import importlib.util
import pathlib

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[1]
spec = importlib.util.spec_from_file_location("flood_stream_model", ROOT / "tools" / "flood_stream_model.py")
model = importlib.util.module_from_spec(spec)
spec.loader.exec_module(model)


def _case(rng, trial):
    H, W = int(rng.integers(1, 48)), int(rng.integers(1, 48))
    dens = rng.choice([0.0, 0.3, 0.6, 0.9, 1.0])
    mask = rng.random((H, W)) < dens if dens < 1 else np.ones((H, W), bool)
    markers = np.zeros((H, W), np.int64)
    style = trial % 4
    if style == 0:
        markers[rng.random((H, W)) < rng.choice([0.01, 0.1, 0.5, 1.0])] = 1
        markers *= rng.integers(1, 50, size=(H, W))
    elif style == 1:
        for k in range(1, int(rng.integers(1, 30))):
            y, x = rng.integers(0, H), rng.integers(0, W)
            markers[y:y + rng.integers(1, 9), x:x + rng.integers(1, 9)] = k
    elif style == 2:
        markers[::int(rng.integers(1, 5))] = 3
    elif trial % 8 == 3:
        markers[:] = 5
    else:
        markers[rng.integers(0, H), rng.integers(0, W)] = 1
    return mask, markers


def test_marker_phase_model_equals_the_textbook_heap():
    rng = np.random.default_rng(7)
    pushed = jumped = 0
    for trial in range(160):
        mask, markers = _case(rng, trial)
        ref_order, _, ref_pushes = model.ref_order(mask, markers)
        n = int(((markers != 0) & mask).sum())
        got_order, _, got_pushes = model.model_order(mask, markers)
        assert ref_order[:n] == got_order, f"trial {trial}: marker pop order differs"
        assert ref_pushes[:len(got_pushes)] == got_pushes, f"trial {trial}: push order differs"
        pushed += len(got_pushes)
        jumped += sum(1 for a, b in zip(got_order, got_order[1:]) if b < a)
    assert pushed > 1000 and jumped > 1000            # both mechanisms (sinking pushed entries, queue-jumping markers) were exercised

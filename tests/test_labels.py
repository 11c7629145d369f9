"""Label creation of the boundary method (first part): oracle vs vectors from the real reference (CPU),
HIP kernel vs the same vectors (GPU, exact)."""
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
        T.get_label(m, "distance", 20)

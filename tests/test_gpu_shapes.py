"""GPU (MI355X): the network on non-square inputs and odd level sizes (tools/sweep_shapes.py): outputs and parameter
gradients as accurate as the torch-CPU fp32 oracle is against fp64 — covers the kernel-selection rules (halo tile widths
64..4, pixel blocks 32..4, gather fall-backs, per-sample tables, max-pool) away from the benchmark's powers of two."""
import pathlib
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1] / "tools"))


def test_shape_sweep_matches_oracle():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import sweep_shapes
    assert sweep_shapes.main() == 0

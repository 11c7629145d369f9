"""CPU: the C oracle of the post-processing (oracle/postproc_ref.c) against golden vectors produced by the REAL
reference (src/inference/postprocessing.py under scipy 1.7.1 / scikit-image 0.18.3; tools/gen_golden_postproc.py).
Integer / label outputs must be bit-exact."""
import numpy as np
import pytest

from helpers import load_npz
from oracle import postproc_ref as R

DIST = ["distance_generic", "distance_touching", "distance_no_seeds", "distance_small_seeds",
        "distance_quantised_ties", "distance_256", "distance_seed_below_cell"]


def test_gaussian_sigma05_bit_exact():
    fx = load_npz("postproc_gauss_sigma05.npz")
    assert np.array_equal(R.gaussian05(fx["x"]), fx["y"])
    assert np.array_equal(R.gaussian05(fx["x"]), fx["y2d"])
    assert np.array_equal(R.gaussian05(fx["x"][:3, :2]), fx["tiny"])      # lines shorter than the radius (reflect twice)


def test_label8_raster_ids():
    fx = load_npz("postproc_label_order.npz")
    lab, n = R.label8(fx["bin"])
    assert n == fx["label_bool"].max() and np.array_equal(lab, fx["label_bool"])


@pytest.mark.parametrize("name", ["watershed_float", "watershed_ties", "watershed_const"])
def test_watershed_matches_skimage(name):
    fx = load_npz(f"postproc_{name}.npz")
    assert np.array_equal(R.watershed(fx["image"], fx["markers"], fx["mask"]), fx["out"])


@pytest.mark.parametrize("name", DIST)
def test_distance_postprocessing_bit_exact(name):
    fx = load_npz(f"postproc_{name}.npz")
    for j, (th_cell, th_seed) in enumerate(fx["th"]):
        got, margin = R.distance_postprocessing(fx["border"][..., None], fx["cell"][..., None], th_seed=th_seed,
                                                th_cell=th_cell, return_margin=True)
        want = fx[f"labels_hw1_{j}"]
        assert got.dtype == np.uint16 and got.shape == want.shape
        assert np.array_equal(got, want), f"{name}[{j}] column-major ids; tan margin {margin} ulp"
        got2 = R.distance_postprocessing(fx["border"], fx["cell"], th_seed=th_seed, th_cell=th_cell)
        assert np.array_equal(got2, fx[f"labels_2d_{j}"]), f"{name}[{j}] raster ids"


@pytest.mark.parametrize("name", ["boundary_generic", "boundary_touching"])
def test_boundary_postprocessing_bit_exact(name):
    fx = load_npz(f"postproc_{name}.npz")
    assert np.array_equal(R.boundary_postprocessing(fx["probs"]), fx["labels"])

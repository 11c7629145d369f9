"""GPU (MI355X): HIP post-processing through the C ABI — bit-exact against (a) golden vectors produced by the real
reference (scipy/scikit-image) and (b) the C oracle on larger seeded frames; plus size-independent properties."""
import numpy as np
import pytest
import torch

from helpers import load_npz

pytestmark = pytest.mark.gpu
DIST = ["distance_generic", "distance_touching", "distance_no_seeds", "distance_small_seeds",
        "distance_quantised_ties", "distance_256", "distance_seed_below_cell"]


@pytest.fixture(scope="module")
def pp():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from microbeseg_amd.inference import postprocessing
    return postprocessing


@pytest.mark.parametrize("name", DIST)
def test_distance_postprocessing_matches_reference(name, pp):
    fx = load_npz(f"postproc_{name}.npz")
    for j, (th_cell, th_seed) in enumerate(fx["th"]):
        got = pp.distance_postprocessing(border_prediction=fx["border"][..., None], cell_prediction=fx["cell"][..., None],
                                         th_seed=th_seed, th_cell=th_cell)
        want = fx[f"labels_hw1_{j}"]
        assert got.dtype == np.uint16 and got.shape == want.shape
        assert np.array_equal(got, want), f"{name}[{j}]: {(got != want).sum()} px differ"
        got2 = pp.distance_postprocessing(fx["border"], fx["cell"], th_seed=th_seed, th_cell=th_cell)
        assert np.array_equal(got2, fx[f"labels_2d_{j}"]), f"{name}[{j}] (2-D input, raster ids)"


@pytest.fixture(params=[1, 0], ids=["marker_phase_closed_form", "marker_phase_heap_replay"])
def const_stream(request):
    """the marker phase of the constant-image flood both ways (mseg_postproc_set_const_stream): the closed form of the heap's
    behaviour with equal keys (default) and the replay of the heap itself"""
    from microbeseg_amd import _lib
    lib = _lib.load()
    assert lib.mseg_postproc_set_const_stream(request.param) == 0
    yield request.param
    lib.mseg_postproc_set_const_stream(1)


@pytest.mark.parametrize("name", ["boundary_generic", "boundary_touching"])
def test_boundary_postprocessing_matches_reference(name, pp, const_stream):
    fx = load_npz(f"postproc_{name}.npz")
    got = pp.boundary_postprocessing(fx["probs"])
    assert np.array_equal(got, fx["labels"]), f"{(got != fx['labels']).sum()} px differ"


def _synthetic(seed, H, W, n_cells):
    rng = np.random.Generator(np.random.PCG64(seed))
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    cell = np.zeros((H, W), np.float32)
    border = np.zeros((H, W), np.float32)
    for _ in range(n_cells):
        cy, cx, r = rng.uniform(0, H), rng.uniform(0, W), rng.uniform(5, 14)
        d = np.sqrt((yy - cy) ** 2 + (xx - cx) ** 2)
        blob = np.clip(1 - d / r, 0, 1).astype(np.float32)
        border = np.maximum(border, np.minimum(cell, blob) * 1.5)   # overlap zones act as borders
        cell = np.maximum(cell, blob)
    cell += rng.normal(0, 0.01, cell.shape).astype(np.float32)
    border = np.clip(border + rng.normal(0, 0.01, cell.shape).astype(np.float32), 0, 1).astype(np.float32)
    return cell, border


@pytest.mark.parametrize("H,W,n", [(200, 300, 60), (512, 512, 300), (1024, 768, 900)])
def test_distance_postprocessing_matches_oracle_large(H, W, n, pp):
    from oracle import postproc_ref as R
    cell, border = _synthetic(7 + H, H, W, n)
    for th_cell, th_seed in ((0.10, 0.45), (0.05, 0.35)):
        want, margin = R.distance_postprocessing(border[..., None], cell[..., None], th_seed, th_cell, return_margin=True)
        got = pp.distance_postprocessing(border[..., None], cell[..., None], th_seed=th_seed, th_cell=th_cell)
        assert np.array_equal(got, want), f"{(got != want).sum()} px differ (tan margin {margin} ulp)"
        # properties: labels live inside the mask, ids are 1..N without gaps, idempotent call
        ids = np.unique(got)
        assert ids[0] == 0 and np.array_equal(ids[1:], np.arange(1, len(ids)))
        assert np.array_equal(pp.distance_postprocessing(border[..., None], cell[..., None], th_seed=th_seed,
                                                         th_cell=th_cell), got)


def test_device_resident_call_and_status(pp):
    from oracle import postproc_ref as R
    cell, border = _synthetic(3, 256, 256, 50)
    c, b = torch.from_numpy(cell).cuda(), torch.from_numpy(border).cuda()
    labels, n_inst, status = pp.distance_postprocessing_device(b, c, 0.45, 0.10, col_major_ids=True)
    want = R.distance_postprocessing(border[..., None], cell[..., None], 0.45, 0.10)
    assert np.array_equal(labels.cpu().numpy().view(np.uint16), want)
    assert int(n_inst) == int(want.max())
    assert int(status) & 1 == 0, "tie-free float data must stay on the parallel per-component flood"


def test_quantised_frame_takes_exact_serial_path(pp):
    """Heavily quantised values force age-0 ties inside components -> taint -> device-side exact serial redo."""
    from oracle import postproc_ref as R
    cell, border = _synthetic(11, 160, 160, 30)
    cell = (np.round(cell * 8) / 8).astype(np.float32)
    c, b = torch.from_numpy(cell).cuda(), torch.from_numpy(border).cuda()
    labels, _, status = pp.distance_postprocessing_device(b, c, 0.45, 0.10, col_major_ids=True)
    want = R.distance_postprocessing(border[..., None], cell[..., None], 0.45, 0.10)
    assert np.array_equal(labels.cpu().numpy().view(np.uint16), want)


def test_empty_and_full_frames(pp):
    z = np.zeros((64, 96), np.float32)
    assert not pp.distance_postprocessing(z[..., None], z[..., None], th_seed=0.45, th_cell=0.10).any()
    one = np.ones((64, 96), np.float32)
    out = pp.distance_postprocessing(z[..., None], one[..., None], th_seed=0.45, th_cell=0.10)
    assert (out == 1).all()          # one seed covering everything -> a single instance


@pytest.mark.parametrize("levels", [1 << 14, 1 << 11, 1 << 8, 1 << 5])
def test_tie_stress_fast_path_stays_exact(levels, pp):
    """Randomised frames whose values are quantised so that equal-valued initial markers DO occur inside components.
    Whatever path the device takes (parallel per-component floods when the tie rules prove them safe, exact serial
    redo otherwise) the labels must equal the oracle's; with light quantisation most frames must stay parallel —
    i.e. the tie rules are exercised on frames that really contain ties."""
    from oracle import postproc_ref as R
    from microbeseg_amd.utils import synth
    parallel_frames = tied_parallel = 0
    n_frames = 24
    for s in range(n_frames):
        rng = np.random.Generator(np.random.PCG64(1000 * levels + s))
        H, W = int(rng.integers(90, 150)), int(rng.integers(90, 150))
        cell, border = synth.synth_prediction_maps(rng, H, W, int(rng.integers(15, 45)), rmin=5, rmax=12)
        cell = (np.round(cell * levels) / levels).astype(np.float32)
        c, b = torch.from_numpy(cell).cuda(), torch.from_numpy(border).cuda()
        labels, _, status = pp.distance_postprocessing_device(b, c, 0.45, 0.10, col_major_ids=True)
        want = R.distance_postprocessing(border[..., None], cell[..., None], 0.45, 0.10)
        got = labels.cpu().numpy().view(np.uint16)
        assert np.array_equal(got, want), f"levels {levels} frame {s}: {(got != want).sum()} px differ, status {int(status)}"
        if int(status) & 1 == 0:
            parallel_frames += 1
            seeds = cell[(cell > 0.45)]
            tied_parallel += int(len(np.unique(seeds)) < len(seeds))
    if levels >= (1 << 11):
        assert parallel_frames >= n_frames // 2, f"only {parallel_frames}/{n_frames} frames stayed on the parallel path"
        assert tied_parallel > 0, "no frame with tied seed values went through the parallel path"


@pytest.mark.parametrize("H,W,n,seed", [(200, 260, 60, 1), (384, 512, 260, 2), (97, 131, 25, 3), (700, 900, 900, 4),
                                        (64, 2048, 120, 5), (33, 17, 3, 6)])
def test_boundary_postprocessing_matches_oracle_random(H, W, n, seed, pp, const_stream):
    """Boundary method on random frames with leaky boundaries (several seeds per mask component, so labels meet inside
    components and every tie is decided by the age order): marker phase (closed form / heap replay) + ordered parallel BFS
    vs the C oracle."""
    from microbeseg_amd.utils import synth
    from oracle import postproc_ref
    rng = np.random.Generator(np.random.PCG64(900 + seed))
    cell, border = synth.synth_prediction_maps(rng, H, W, n, rmin=5.0, rmax=12.0)
    gaps = rng.uniform(0, 1, (H, W)) < 0.35                      # holes in the boundary class -> merged components
    p1 = np.clip(cell * 2.5, 0, 1) * (1 - np.clip(border * 1.2, 0, 1) * ~gaps)
    p2 = np.clip(border * 1.2, 0, 1) * (cell > 0.02) * ~gaps
    p0 = np.clip(1 - p1 - p2, 0.0, 1)
    probs = np.stack([p0, p1, p2], -1).astype(np.float32)
    probs /= probs.sum(-1, keepdims=True)
    got = pp.boundary_postprocessing(probs)
    want = postproc_ref.boundary_postprocessing(probs)
    assert got.dtype == np.uint16 and np.array_equal(got, want)
    assert want.max() >= 3


@pytest.mark.parametrize("rows,tile_s,tile_l", [(1, -1, -1), (16, 0, 0), (1, 0, 0), (2, 400, 2000), (16, 0, -1)])
def test_component_flood_paths_agree(rows, tile_s, tile_l, pp):
    """The per-component flood (one wavefront per mask component) keeps its queue in LDS with a spill into the workspace
    and stages the component's bounding box in LDS, probing global memory for boxes beyond the tile.  mseg_postproc_tuning
    shrinks the LDS shares so that small test frames drive the spill rows, the large-tile launch and the global-probe
    path; labels must stay bit-identical to the oracle on tie-free and on quantised (tied) frames alike."""
    from oracle import postproc_ref as R
    from microbeseg_amd import _lib
    from microbeseg_amd.utils import synth
    lib = _lib.load()
    assert lib.mseg_postproc_tuning(0, -1, -1) != 0 and lib.mseg_postproc_tuning(17, -1, -1) != 0
    assert lib.mseg_postproc_tuning(rows, tile_s, tile_l) == 0
    try:
        for s, levels in enumerate((0, 0, 1 << 12, 1 << 9)):
            rng = np.random.Generator(np.random.PCG64(4242 + s))
            H, W = int(rng.integers(150, 260)), int(rng.integers(150, 260))
            cell, border = synth.synth_prediction_maps(rng, H, W, int(rng.integers(80, 160)), rmin=5, rmax=13)
            if levels:
                cell = (np.round(cell * levels) / levels).astype(np.float32)
            for th_cell, th_seed in ((0.10, 0.45), (0.02, 0.30)):      # the low cell threshold merges cells into big components
                c, b = torch.from_numpy(cell).cuda(), torch.from_numpy(border).cuda()
                labels, _, status = pp.distance_postprocessing_device(b, c, th_seed, th_cell, col_major_ids=True)
                want = R.distance_postprocessing(border[..., None], cell[..., None], th_seed, th_cell)
                got = labels.cpu().numpy().view(np.uint16)
                assert np.array_equal(got, want), f"frame {s} th {th_cell}: {(got != want).sum()} px differ, status {int(status)}"
                if not levels:
                    assert int(status) & 1 == 0, "tie-free data must stay on the per-component flood"
    finally:
        assert lib.mseg_postproc_tuning(-1, -1, -1) == 0


def test_boundary_batch_equals_single_frames():
    """mseg_boundary_postprocess_pre / _flood_batch / _post (the floods of up to 8 frames in ONE launch, one workgroup per
    frame) against mseg_boundary_postprocess frame by frame: labels, instance counts and status words identical, for
    batches of 1, 3 and 8 DIFFERENT frames (touching cells, empty frame, one-cell frame), twice on the same workspaces."""
    import torch
    from microbeseg_amd.inference import postprocessing as pp
    from microbeseg_amd.utils import synth
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    rng = np.random.Generator(np.random.PCG64(321))
    H, W = 300, 420
    frames = []
    for t in range(8):
        ncell = [30, 0, 1, 80, 55, 12, 140, 7][t]
        if ncell:
            cell, border = synth.synth_prediction_maps(rng, H, W, ncell, rmin=4.0, rmax=12.0)
        else:
            cell, border = np.zeros((H, W), np.float32), np.zeros((H, W), np.float32)
        p1 = np.clip(cell * 2.0, 0, 1) * (1 - np.clip(border * 1.2, 0, 1))
        p2 = np.clip(border * 1.2, 0, 1) * (cell > 0.02)
        p0 = np.clip(1 - p1 - p2, 0.0, 1)
        probs = np.stack([p0, p1, p2], -1).astype(np.float32)
        frames.append(torch.from_numpy(probs / np.maximum(probs.sum(-1, keepdims=True), 1e-9)).cuda())
    single = [pp.boundary_postprocessing_device(f) for f in frames]
    single = [(l.cpu().numpy().copy(), int(n), int(s)) for l, n, s in single]
    assert max(n for _, n, _ in single) > 50
    for B in (1, 3, 8, 8):
        for start in range(0, 8, B):
            got = pp.boundary_postprocessing_batch_device(frames[start:start + B], first_slot=2)
            for k, (l, n, s) in enumerate(got):
                want = single[start + k]
                assert np.array_equal(l.cpu().numpy(), want[0]), (B, start + k)
                assert (int(n), int(s)) == want[1:], (B, start + k)

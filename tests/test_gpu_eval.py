"""GPU (MI355X): evaluation path through the C ABI — relabelling / AJI+ against vectors from the real
reference, the threshold sweep against single post-processing calls, and EvalWorker end to end against the CPU oracle."""
import json
import pathlib
import sys

import numpy as np
import pytest
import torch

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from oracle import eval_ref, postproc_ref  # noqa: E402

pytestmark = pytest.mark.gpu
G = np.load(ROOT / "tests" / "golden" / "eval_aji.npz")
CASES = sorted({k.split("_")[0] for k in G.files if k.startswith("c")})


@pytest.fixture(scope="module")
def su():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from microbeseg_amd.evaluation import stats_utils
    return stats_utils


@pytest.mark.parametrize("c", CASES)
def test_relabel_and_aji_match_reference(su, c):
    gt, pred = G[f"{c}_gt"], G[f"{c}_pred"]
    gl, ng = su.relabel_device(gt)
    pl, npd = su.relabel_device(pred)
    assert np.array_equal(gl.cpu().numpy(), G[f"{c}_gt_lab"]) and ng == int(G[f"{c}_gt_lab"].max())
    assert np.array_equal(pl.cpu().numpy(), G[f"{c}_pred_lab"]) and npd == int(G[f"{c}_pred_lab"].max())
    want = float(G[f"{c}_aji"])
    assert su.aji_plus_masks(pred, gt) == pytest.approx(want, rel=1e-12, abs=1e-15)
    if want > 0:
        assert su.get_fast_aji_plus(G[f"{c}_gt_lab"], G[f"{c}_pred_lab"]) == pytest.approx(want, rel=1e-12)


def test_relabel_known_answers(su):
    lab, k = su.relabel_device(G["label_in"].astype(np.uint16), border_width=0)
    assert np.array_equal(lab.cpu().numpy(), G["label_out"]) and k == int(G["label_out"].max())
    for bw, key in ((10, "bc_w10"), (3, "bc_w3")):
        lab, _ = su.relabel_device(G["bc_in"], border_width=bw)
        assert np.array_equal(lab.cpu().numpy(), eval_ref.label_image(G[key]))
    # border wider than the frame: nothing is visible in the (empty) field of interest
    lab, k = su.relabel_device(G["bc_in"][:15, :15].copy(), border_width=10)
    assert k == 0 and not lab.any()


def test_pair_counts_random(su):
    rng = np.random.default_rng(5)
    t = rng.integers(0, 40, (150, 170)).astype(np.int32)
    p = rng.integers(0, 55, (150, 170)).astype(np.int32)
    at, ap, inter = su.pair_counts_device(torch.from_numpy(t).cuda(), torch.from_numpy(p).cuda(), 39, 54)
    wt, wp, wi = eval_ref.pair_counts(t, p)
    assert np.array_equal(at[1:], wt[1:]) and np.array_equal(ap[1:], wp[1:]) and np.array_equal(inter[1:, 1:], wi[1:, 1:])


def test_threshold_sweep_equals_single_calls():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from microbeseg_amd.inference import postprocessing as pp
    from microbeseg_amd.utils import synth
    rng = np.random.Generator(np.random.PCG64(99))
    cell, border = synth.synth_prediction_maps(rng, 192, 256, 40, rmin=5.0, rmax=12.0)
    c, b = torch.from_numpy(cell).cuda(), torch.from_numpy(border).cuda()
    ths = [(tc, ts) for tc in (0.05, 0.075, 0.10, 0.125) for ts in (0.35, 0.45)]
    labels, n_inst, status = pp.distance_postprocessing_sweep_device(b, c, ths)
    got = labels.cpu().numpy().view(np.uint16)
    for k, (tc, ts) in enumerate(ths):
        one, n1, _ = pp.distance_postprocessing_device(b, c, th_seed=ts, th_cell=tc)
        assert np.array_equal(got[k], one.cpu().numpy().view(np.uint16)) and int(n_inst[k]) == int(n1)
        want = postproc_ref.distance_postprocessing(border[..., None], cell[..., None], ts, tc)
        assert np.array_equal(got[k], want)
    assert len({got[k].tobytes() for k in range(len(ths))}) > 1      # the grid does change the masks on this frame


def test_eval_worker_end_to_end(tmp_path):
    """Synthetic data set -> TrainWorker (tiny DU-Net, a few epochs) -> EvalWorker; the masks it keeps, the chosen
    thresholds and the mean AJI+ equal the CPU oracle chain run on the same network outputs."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from microbeseg_amd.evaluation.eval import EvalWorker
    from microbeseg_amd.training.train import TrainWorker
    from microbeseg_amd.utils import synth, tiffio
    from microbeseg_amd.utils.unets import build_unet, get_weights
    from microbeseg_amd.utils.utils import zero_pad_model_input
    dev = torch.device("cuda:0")
    data = synth.write_training_set(tmp_path / "set", 48, 8, size=128, seed=1234, label_types=("distance",))
    (data / "test").mkdir()
    rng = np.random.Generator(np.random.PCG64(4321))
    n_img = 5
    for i in range(n_img):
        c = synth.synth_crop(rng, 128)
        tiffio.imwrite(str(data / "test" / f"img_{i:03d}.tif"), c["img"])
        tiffio.imwrite(str(data / "test" / f"mask_{i:03d}.tif"), c["mask"])
    models = tmp_path / "models" / "runA"
    models.mkdir(parents=True)
    tw = TrainWorker()
    tw.num_workers = 0
    tw.start_training(data, models, "distance", 1, "adam", 8, dev, 1, False, filters=[16, 64], max_epochs=80)
    model = models / "distance_model_01.pth"
    cfg = json.load(open(models / "distance_model_01.json"))
    results = tmp_path / "results" / "set"
    results.parent.mkdir()
    w = EvalWorker()
    w.num_workers = 0
    w.start_evaluation(data, results, [model], 2, dev, 1, False, start_message="go")
    out_dir = results / "runA_distance_model_01"
    assert (out_dir / "scores.csv").is_file() and (out_dir / "test_set.zip").is_file()
    summary = open(results.parent / "set.csv").read().strip().splitlines()
    assert summary[0].split(",")[:3] == ["model", "th_cell", "th_seed"] and len(summary) == 2
    # oracle chain on the same network outputs
    arch = cfg["architecture"]
    net = build_unet(arch[0], arch[2], arch[1], arch[3], dev, 1, ch_out=1, filters=arch[4])
    net = get_weights(net=net, weights=str(model), num_gpus=1, device=dev)
    net.eval()
    ths = [(tc, ts) for tc in (0.05, 0.075, 0.10, 0.125) for ts in (0.35, 0.45)]
    per_th = {th: [] for th in ths}
    masks = {th: {} for th in ths}
    for i in range(n_img):
        img = tiffio.imread(str(data / "test" / f"img_{i:03d}.tif"))
        gt = tiffio.imread(str(data / "test" / f"mask_{i:03d}.tif"))
        x = 2 * (img.astype(np.float32) - img.min()) / (img.max() - img.min()) - 1
        x, pads = zero_pad_model_input(x, pad_val=np.min(x))
        with torch.no_grad():
            border, cell = net(torch.from_numpy(np.ascontiguousarray(x[None, None])).to(dev))
        border = border[0, 0, pads[0]:, pads[1]:].cpu().numpy()
        cell = cell[0, 0, pads[0]:, pads[1]:].cpu().numpy()
        for th in ths:
            m = postproc_ref.distance_postprocessing(border[..., None], cell[..., None], th[1], th[0])
            masks[th][i] = m
            per_th[th].append(eval_ref.score_pair(m, gt))
    means = {th: float(np.mean(v)) for th, v in per_th.items()}
    best = None
    for th in sorted(ths, key=lambda t: "{}_{}".format(t[0], t[1])):      # directory order of the worker
        if best is None or means[th] > means[best]:
            best = th
    row = summary[1].split(",")
    assert float(row[1]) == best[0] and float(row[2]) == best[1]
    assert float(row[3]) == pytest.approx(means[best], rel=1e-9, abs=1e-12)
    print('AJI+ per threshold pair:', means)
    assert means[best] > 0.2          # the tiny net did learn something: the comparison is not vacuous
    for i in range(n_img):
        kept = tiffio.imread(str(out_dir / f"mask_{i:03d}.tif"))
        assert np.array_equal(kept, masks[best][i])


def test_other_metrics_match_reference():
    """get_fast_aji / get_fast_pq / get_fast_dice_2 / get_dice_1 / get_dice_2 / remap_label / pair_coordinates of the
    reference module (stats_utils.py:16-95, 183-437) on the device's pair statistics vs vectors from the real functions."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from src.evaluation import stats_utils as S
    GM = np.load(ROOT / "tests" / "golden" / "eval_metrics.npz")
    for c in range(5):
        gl, pl = G[f"c{c}_gt_lab"], G[f"c{c}_pred_lab"]
        assert abs(S.get_fast_aji(gl, pl) - float(GM[f"c{c}_aji"])) < 1e-12
        for tag, thr in (("pq50", 0.5), ("pq30", 0.3)):
            (dq, sq, pq), (pt, pp, ut, up) = S.get_fast_pq(gl, pl, match_iou=thr)
            assert np.allclose([dq, sq, pq], GM[f"c{c}_{tag}"], rtol=0, atol=1e-12)
            assert np.array_equal(np.asarray(pt), GM[f"c{c}_{tag}_pt"]) and np.array_equal(np.asarray(pp), GM[f"c{c}_{tag}_pp"])
            assert np.array_equal(np.asarray(ut, np.int64), GM[f"c{c}_{tag}_ut"])
            assert np.array_equal(np.asarray(up, np.int64), GM[f"c{c}_{tag}_up"])
        assert abs(S.get_fast_dice_2(gl, pl) - float(GM[f"c{c}_dice2"])) < 1e-12
        assert abs(S.get_dice_2(gl, pl) - float(GM[f"c{c}_dice2_slow"])) < 1e-12
        assert abs(S.get_dice_1(gl, pl) - float(GM[f"c{c}_dice1"])) < 1e-12
        assert np.array_equal(S.remap_label(G[f"c{c}_pred"]), GM[f"c{c}_remap"])
        assert np.array_equal(S.remap_label(G[f"c{c}_pred"], by_size=True), GM[f"c{c}_remap_size"])
    pairing, ua, ub = S.pair_coordinates(GM["pc_A"], GM["pc_B"], 4.0)
    assert np.array_equal(pairing, GM["pc_pairing"]) and np.array_equal(ua, GM["pc_ua"]) and np.array_equal(ub, GM["pc_ub"])

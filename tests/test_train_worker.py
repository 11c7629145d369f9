"""Training driver (TrainWorker mirror) end to end on BASELINE configs[0]-shaped work: 2-level, 8-filter net,
synthetic 256x256 crops, batch 2, one epoch -> checkpoint / json / loss-log / zip contract.
  * CPU: host logic only, with the oracle network injected (tests may use the oracle; the product never does).
  * GPU: the real HIP path, then the saved checkpoint is re-loaded by the CPU oracle and by the inference driver."""
import json
import zipfile

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import unet_ref


class OracleNet(nn.Module):
    """CPU stand-in with the product's parameter tree; forward = oracle/unet_ref.py (test infrastructure)."""

    def __init__(self, holder, ut, act, norm, filters):
        super().__init__()
        self.holder, self.cfg = holder, (ut, act, norm, filters)

    def forward(self, x):
        sd = dict(self.holder.named_parameters())
        sd.update(dict(self.holder.named_buffers()))
        ut, act, norm, filters = self.cfg
        return unet_ref.unet_forward(sd, x, ut, act, norm, filters, training=self.training,
                                     update_running_stats=self.training)

    def state_dict(self, *a, **k):
        return self.holder.state_dict(*a, **k)

    def load_state_dict(self, sd, *a, **k):
        return self.holder.load_state_dict(sd, *a, **k)


def _check_outputs(models, label_type, n_epochs):
    pth = models / f"{label_type}_model_01.pth"
    sd = torch.load(pth, map_location="cpu")
    assert "encoderConv.0.conv.0.weight" in sd and sd["encoderConv.0.conv.0.weight"].shape == (8, 1, 3, 3)
    cfg = json.load(open(models / f"{label_type}_model_01.json"))
    assert cfg["architecture"][:4] == ["DU" if label_type == "distance" else "U", "conv", "relu", "bn"]
    assert cfg["architecture"][4] == [8, 16] and cfg["label_type"] == label_type and cfg["batch_size"] == 2
    assert cfg["loss"] == ("smooth_l1" if label_type == "distance" else "ce_dice")
    assert cfg["max_epochs"] == n_epochs and cfg["trained_epochs"] == n_epochs and "training_time" in cfg
    log = (models / f"{label_type}_model_01_loss.txt").read_text().splitlines()
    assert log[0] == "# Epoch, training loss, validation loss" and len(log) == 1 + n_epochs
    assert len(log[1].split(",")) == 3
    with zipfile.ZipFile(models / f"{label_type}_model_01_trainset.zip") as z:
        names = z.namelist()
    assert any(n.endswith("train/img_000.tif") for n in names) and not any("/test/" in n for n in names)
    return sd, cfg


@pytest.mark.parametrize("label_type", ["distance", "boundary"])
def test_train_worker_host_logic_cpu(tmp_path, monkeypatch, label_type):
    from microbeseg_amd.training import train as T
    from microbeseg_amd.utils import synth
    from microbeseg_amd.utils.unets import build_unet as real_build
    data = synth.write_training_set(tmp_path / "set", 6, 4, size=64, seed=5)

    def fake_build(unet_type, act_fun, pool_method, normalization, device, num_gpus, ch_in=1, ch_out=1,
                   filters=(64, 1024)):
        holder = real_build(unet_type, act_fun, pool_method, normalization, "cpu", 1, ch_in, ch_out, tuple(filters))
        return OracleNet(holder, unet_type, act_fun, normalization, tuple(filters))

    def fake_loss(loss_function, label_type):
        if label_type == "distance":
            return {"border": unet_ref.regression_loss, "cell": unet_ref.regression_loss}
        return unet_ref.ce_dice

    monkeypatch.setattr(T, "build_unet", fake_build)
    monkeypatch.setattr(T, "get_loss", fake_loss)
    # the product's Adam is the fused HIP optimizer (no CPU fallback); the host-logic test injects torch's
    monkeypatch.setitem(T.OPTIMIZER_RECIPES["adam"], "make",
                        lambda ps, lr: torch.optim.Adam(ps, lr=lr, betas=(0.9, 0.999), eps=1e-8, amsgrad=True))
    w = T.TrainWorker()
    w.augment = False      # the device augmentation needs the GPU; this test exercises the host logic only
    msgs, prog = [], []
    w.text_output.connect(msgs.append)
    w.progress.connect(prog.append)
    models = tmp_path / "models"
    models.mkdir()
    w.start_training(data, models, label_type, 1, "adam", 2, torch.device("cpu"), 1, False, filters=[8, 16],
                     max_epochs=2)
    _check_outputs(models, label_type, 2)
    assert any("--> save" in m for m in msgs) and prog[-1] == 100
    assert any(m.startswith("Train/validate on 6/4 images") for m in msgs)


def test_stop_request_removes_checkpoint(tmp_path, monkeypatch):
    from microbeseg_amd.training import train as T
    w = T.TrainWorker()
    w.stop_training_process()
    assert w.stop_training
    fin = []
    w.finished.connect(lambda: fin.append(1))
    (tmp_path / "empty").mkdir()
    w.start_training(tmp_path / "empty", tmp_path, "distance", 1, "adam", 2, torch.device("cpu"), 1)
    assert fin == [1]                      # empty training set -> finished immediately (train.py:150-154)


@pytest.mark.gpu
@pytest.mark.parametrize("label_type,optimizer", [("distance", "adam"), ("boundary", "adam"), ("distance", "ranger")])
def test_train_worker_on_gpu_configs0(tmp_path, label_type, optimizer):
    """BASELINE configs[0]: 2-level U-Net, 8 base filters, 32 synthetic 256x256 crops, batch 2, 1 epoch — on the HIP
    path; the checkpoint it writes is then consumed by the CPU oracle (state-dict contract) and by InferWorker."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from microbeseg_amd.training.train import TrainWorker
    from microbeseg_amd.inference.infer import InferWorker
    from microbeseg_amd.utils import synth
    data = synth.write_training_set(tmp_path / "set", 32, 8, size=256, seed=1234)
    models = tmp_path / "models"
    models.mkdir()
    w = TrainWorker()
    w.num_workers = 0
    msgs = []
    w.text_output.connect(msgs.append)
    w.start_training(data, models, label_type, 1, optimizer, 2, torch.device("cuda:0"), 1, False, filters=[8, 16],
                     max_epochs=10 if optimizer == "ranger" else 1)
    pth = models / f"{label_type}_model_01.pth"
    sd = torch.load(pth, map_location="cpu")
    cfg = json.load(open(models / f"{label_type}_model_01.json"))
    assert cfg["optimizer"] == optimizer and cfg["architecture"][2] == ("mish" if optimizer == "ranger" else "relu")
    if optimizer == "ranger":
        assert "trained_epochs_run2" in cfg and any("2nd run" in m for m in msgs)
    # the checkpoint drives the CPU oracle and the HIP inference path to the same prediction
    x = torch.rand(1, 1, 64, 64) * 2 - 1
    ut, act, norm, filters = cfg["architecture"][0], cfg["architecture"][2], cfg["architecture"][3], tuple(cfg["architecture"][4])
    with torch.no_grad():
        ref = unet_ref.unet_forward(sd, x, ut, act, norm, filters, training=False)
    iw = InferWorker(model=str(models / f"{label_type}_model_01"), device="cuda:0")
    with torch.no_grad():
        got = iw.net(x.cuda())
    ref = ref if isinstance(ref, tuple) else (ref,)
    got = got if isinstance(got, tuple) else (got,)
    for r, g in zip(ref, got):
        assert (g.cpu() - r).abs().max().item() <= 1e-4 * max(1.0, r.abs().max().item())
    frame = (np.random.default_rng(0).random((100, 130)) * 4000).astype(np.uint16)
    masks = iw.infer_stack(frame[None])
    assert masks.shape == (1, 100, 130) and masks.dtype == np.uint16
    # a stack is pipelined over two streams (distance models): every mask equals the frame-by-frame result
    rng = np.random.default_rng(1)
    stack = (rng.random((5, 100, 130)) * 4000).astype(np.uint16)
    masks = iw.infer_stack(stack)
    from microbeseg_amd.utils.utils import zero_pad_model_input
    for t in range(len(stack)):
        f = np.copy(stack[t])
        fmin, fmax = np.min(f), np.max(f)
        fp, pads = zero_pad_model_input(f, pad_val=fmin)
        assert np.array_equal(masks[t], iw.inference(fp, fmin, fmax, pads))


@pytest.mark.gpu
def test_train_worker_bf16_precision(tmp_path):
    """BASELINE configs[2] through the worker: ``worker.precision = 'bf16'`` trains with bf16 matrix-core inputs (fp32
    accumulate / storage / statistics), records it in the run's .json, restores the engine's mode afterwards, and the
    training loss falls like the fp32 run's on the same data and seed."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from microbeseg_amd import engine
    from microbeseg_amd.training.train import TrainWorker
    from microbeseg_amd.utils import synth
    data = synth.write_training_set(tmp_path / "set", 16, 4, size=128, seed=77)
    losses = {}
    for prec in ("fp32", "bf16"):
        models = tmp_path / f"models_{prec}"
        models.mkdir()
        torch.manual_seed(3)
        np.random.seed(3)
        import random
        random.seed(3)
        w = TrainWorker()
        w.num_workers = 0
        w.augment = False
        w.precision = prec
        w.start_training(data, models, "distance", 1, "adam", 4, torch.device("cuda:0"), 1, False, filters=[64, 128],
                         max_epochs=6)
        assert engine.get_precision() == "fp32"
        cfg = json.load(open(models / "distance_model_01.json"))
        assert cfg.get("precision", "fp32") == prec
        log = np.loadtxt(models / "distance_model_01_loss.txt", ndmin=2, delimiter=",")
        losses[prec] = log[:, 1]                                       # training loss per epoch
    assert losses["bf16"][-1] < 0.7 * losses["bf16"][0]
    assert abs(losses["bf16"][-1] - losses["fp32"][-1]) < 0.15 * losses["fp32"][-1]


@pytest.mark.gpu
def test_infer_worker_bf16_option(tmp_path):
    """Opt-in bf16 inference: a trained distance model segments a synthetic frame almost like the fp32 run (AJI+ of the
    two masks >= 0.9 — not bit-identical by construction), the default stays fp32 and the engine's mode is restored."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from microbeseg_amd import engine
    from microbeseg_amd.training.train import TrainWorker
    from microbeseg_amd.inference.infer import InferWorker
    from microbeseg_amd.evaluation.stats_utils import aji_plus_masks
    from microbeseg_amd.utils import synth
    data = synth.write_training_set(tmp_path / "set", 24, 6, size=128, seed=21)
    models = tmp_path / "models"
    models.mkdir()
    torch.manual_seed(5)
    w = TrainWorker()
    w.num_workers = 0
    w.augment = False
    w.start_training(data, models, "distance", 1, "adam", 4, torch.device("cuda:0"), 1, False, filters=[64, 128],
                     max_epochs=40)
    rng = np.random.Generator(np.random.PCG64(99))
    frames = np.stack([synth.synth_crop(rng, 192)["img"] for _ in range(3)])
    iw = InferWorker(model=str(models / "distance_model_01"), device="cuda:0")
    assert iw.precision == "fp32"
    m32 = iw.infer_stack(frames)
    iw.precision = "bf16"
    m16 = iw.infer_stack(frames)
    assert engine.get_precision() == "fp32"
    assert m16.shape == m32.shape and m16.dtype == np.uint16
    assert m32.max() > 3                                   # the model segments something
    for a, b in zip(m32, m16):
        assert aji_plus_masks(a, b) >= 0.9

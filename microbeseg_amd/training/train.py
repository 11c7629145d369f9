"""Training driver of the hot path (host side; every FLOP runs in libmseg_hip).

Drop-in for ``src/training/train.py`` of the reference: ``TrainWorker.start_training`` (:124-304), ``TrainWorker.train``
(:316-576), ``CreateLabelsWorker.create_labels`` (:33-112), ``get_max_epochs`` (:579-606), ``seed_worker`` (:609-620) keep
their names, arguments, Qt signals, user-visible messages and output files (``<run>.pth`` = state dict of the bare module,
``<run>.json``, ``<run>_loss.txt``, ``<run>_trainset.zip``; SURVEY.md Appendix C / D are the behaviour spec this file was
written from).  The structure is this build's own:

  ``OPTIMIZER_RECIPES``  the three optimisation regimes (Adam, Ranger, Ranger fine-tune) as data
  ``MemoryLadder``       the out-of-memory back-off (batch 8 -> 4, filters [32,512] -> [32,256]) as a list of rungs
  ``BestCheckpoint``     best-validation bookkeeping: save on strict improvement, count epochs without one
  ``ShardPlan``          which crops a rank sees in a phase (exact ``nn.DataParallel`` scatter for validation)
  ``_Feeder``            host -> HBM + device-side augmentation of one batch
  ``TrainWorker``        ties them together, emits the reference's signals / messages

Additive extensions: Qt is optional (``utils.qt_shim``); ``start_training(..., filters=None, max_epochs=None)``;
``TrainWorker.precision`` ("bf16": BASELINE configs[2]); with ``num_gpus > 1`` under torch.distributed ``batch_size`` is
the GLOBAL batch exactly as under the reference's ``nn.DataParallel`` (each rank takes ``batch_size // world`` crops of
it), validation runs without duplicated crops on rank 0's BatchNorm statistics, epoch losses are global so that every
rank takes the same save / early-stop decisions, and rank 0 writes the files.
"""
import gc
import math
import os
import random
import time
import zipfile
from multiprocessing import cpu_count
from shutil import rmtree

import numpy as np
import torch
from torch.optim.lr_scheduler import CosineAnnealingLR, ReduceLROnPlateau

from .. import parallel
from ..utils import tiffio as tiff
from ..utils.qt_shim import QCoreApplication, QObject, pyqtSignal, pyqtSlot
from ..utils.unets import build_unet, get_weights
from ..utils.utils import unique_path, write_train_info
from .losses import get_loss
from .optim import FusedAdam
from .ranger2020 import Ranger
from .train_data_representations import get_label, max_major_axis_length
from .training_dataset import TrainingDataset, augmentors

PHASES = ('train', 'val')


# ---- label creation ---------------------------------------------------------------------------------------------------
class CreateLabelsWorker(QObject):
    """ Worker class for label creation: writes the training representation of every ``mask*.tif`` in
    ``path/{train,val}`` next to it (``cell_dist_*`` / ``neighbor_dist_*`` for 'distance', ``<label_type>_*`` else). """
    finished = pyqtSignal()
    progress = pyqtSignal(int)
    text_output = pyqtSignal(str)
    stop_label_creation = False

    def _quit(self, percent):
        self.progress.emit(percent)
        self.finished.emit()

    def create_labels(self, path, label_type):
        if not any(path.glob('*')):                     # the export was stopped and its folders removed
            return self._quit(0)
        self.text_output.emit('Create labels')
        masks = {split: sorted((path / split).glob('mask*.tif')) for split in PHASES}
        if min(len(m) for m in masks.values()) < 2:
            self.text_output.emit("The training and the validation set should each contain at least two annotated "
                                  "images! Stop")
            return self._quit(0)
        todo = masks['train'] + masks['val']
        for done, mask_file in enumerate(todo, start=1):
            QCoreApplication.processEvents()            # lets the stop slot run
            if self.stop_label_creation:
                self.text_output.emit("Stop label creation due to user interaction.\nDelete local folder.")
                rmtree(str(path))
                return self._quit(0)
            mask = tiff.imread(str(mask_file))
            # distance labels: the neighbour search radius follows the longest cell of this mask
            radius = int(np.ceil(max_major_axis_length(mask))) if label_type == 'distance' else 0
            label = get_label(mask=mask, label_type=label_type, max_mal=radius)
            tail = mask_file.name.split('mask_')[-1]
            if label_type == 'distance':
                outputs = {'cell_dist_': label[0], 'neighbor_dist_': label[1]}
            else:
                outputs = {label_type + '_': label}
            for prefix, array in outputs.items():
                tiff.imwrite(str(mask_file.parent / (prefix + tail)), array)
            self.progress.emit(int(100 * done / len(todo)))
        return self._quit(100)

    @pyqtSlot()
    def stop_label_creation_process(self):
        self.stop_label_creation = True


# ---- optimisation regimes -------------------------------------------------------------------------
def _adam(params, lr):
    return FusedAdam(params, lr=lr, betas=(0.9, 0.999), eps=1e-08, weight_decay=0, amsgrad=True)


def _ranger(params, lr):
    return Ranger(params, lr=lr, alpha=0.5, k=6, N_sma_threshhold=5, betas=(.95, 0.999), eps=1e-6, weight_decay=0,
                  use_gc=True, gc_conv_only=False, gc_loc=True)


# E = configs['max_epochs'].  epochs: length of the run; stale: epochs without improvement that end it; plateau: the
# scheduler is ReduceLROnPlateau(factor .25) with this patience / floor, else cosine annealing; span: what fraction of an
# iteration's progress bar one epoch of the run fills (Ranger shares the bar between its two runs, 1 : 0.1).
RANGER_LR = 6e-3
OPTIMIZER_RECIPES = {
    'adam': dict(make=_adam, lr=8e-4, epochs=lambda E: E, stale=lambda E: 2 * E // 20 + 5,
                 plateau=lambda E: dict(patience=E // 20, min_lr=3e-6), span=lambda n: n),
    'ranger': dict(make=_ranger, lr=RANGER_LR, epochs=lambda E: E, stale=lambda E: 2 * E // 10 + 5,
                   plateau=lambda E: dict(patience=E // 10, min_lr=0.075 * RANGER_LR), span=lambda n: 1.1 * n),
    'ranger/finetune': dict(make=_ranger, lr=0.09 * RANGER_LR, epochs=lambda E: E // 10, stale=lambda E: E // 10 + 1,
                            plateau=None, cosine=lambda E: dict(T_max=max(1, E // 10), eta_min=3e-5, last_epoch=-1),
                            span=lambda n: 10 * n),
}


class _Regime:
    """optimizer + scheduler + budgets of one call of ``TrainWorker.train``"""

    def __init__(self, name, finetune, params, E):
        key = name + '/finetune' if (finetune and name == 'ranger') else name
        if key not in OPTIMIZER_RECIPES:
            raise Exception('Optimizer not known')
        r = OPTIMIZER_RECIPES[key]
        self.optimizer = r['make'](params, r['lr'])
        self.epochs, self.stale_limit = r['epochs'](E), r['stale'](E)
        self.metric_driven = r['plateau'] is not None
        if self.metric_driven:
            self.scheduler = ReduceLROnPlateau(self.optimizer, mode='min', factor=0.25, **r['plateau'](E))
        else:
            self.scheduler = CosineAnnealingLR(self.optimizer, **r['cosine'](E))
        self._span = r['span'](self.epochs)
        self._graphed = False                        # not decided yet (graphed_step)

    def after_validation(self, val_loss):
        self.scheduler.step(val_loss) if self.metric_driven else self.scheduler.step()

    def graphed_step(self, net, criterion, distance, device, world):
        """the training step of ``_run_phase`` as a replayed hipGraph, or None where that does not apply (several ranks: the
        gradient all-reduce overlaps the backward pass eagerly; Ranger: host-side step logic; CPU).  Built once per run:
        the recorded step stays valid across epochs — the learning rate is read from device memory."""
        if self._graphed is False:
            from .optim import FusedAdam
            self._graphed = None
            if world == 1 and torch.device(device).type == "cuda" and isinstance(self.optimizer, FusedAdam):
                from .graph_step import GraphedTrainStep
                opt = self.optimizer

                def step(img, *labels):
                    opt.zero_grad()
                    if distance:
                        border, cell = net(img)
                        loss = criterion['border'](border, labels[0]) + criterion['cell'](cell, labels[1])
                    else:
                        loss = criterion(net(img), labels[0])
                    loss.backward()
                    opt.step()
                    return loss
                self._graphed = GraphedTrainStep(step, opt, warmup=2)
        return self._graphed

    def percent(self, epochs_done, train_progress):
        return int(100 * epochs_done / self._span * train_progress[0] + 100 * train_progress[1])


# ---- out-of-memory back-off (train.py:276-297 of the reference) -----------------------------------------------------------
def is_out_of_memory(err):
    """The reference treats EVERY RuntimeError of a training run as 'does not fit'.  Only real allocation failures climb
    down the ladder here; a bad shape or a missing library surfaces instead of silently shrinking the model."""
    text = str(err).lower()
    return "out of memory" in text or "hiperroroutofmemory" in text


_is_oom = is_out_of_memory       # name used by earlier rounds / tests


class MemoryLadder:
    """The model shrinks one rung per failure: batch -> 8 -> 4, then filters [32,512], then [32,256], then give up."""
    RUNGS = (
        (lambda s: s.batch_size > 8, dict(batch_size=8), "Model does not fit on RAM/VRAM. Reduce batch size from {b} to 8"),
        (lambda s: s.batch_size > 4, dict(batch_size=4), "Model does not fit on RAM/VRAM. Reduce batch size from {b} to 4"),
        (lambda s: s.filters[0] > 32, dict(filters=[32, 512]), "Model does not fit on RAM/VRAM. Reduce number of kernels"),
        (lambda s: s.filters[-1] == 512, dict(filters=[32, 256]), "Model does not fit on RAM/VRAM. Reduce model depth"),
    )
    GIVE_UP = ("Please, try again with smaller batch size or reduce the crop size (use the export and import "
               "functionalities for this)")

    def __init__(self, batch_size, filters):
        self.batch_size, self.filters = batch_size, list(filters)
        self.exhausted = False

    def step_down(self):
        """apply the first rung that still applies; returns the message for the GUI"""
        for applies, change, message in self.RUNGS:
            if applies(self):
                text = message.format(b=self.batch_size)
                for name, value in change.items():
                    setattr(self, name, value)
                return text
        self.exhausted = True
        return self.GIVE_UP


# ---- best-validation checkpoint policy -------------------------------------------------------------------------------------
class BestCheckpoint:
    """Saves ``<run>.pth`` whenever the validation loss improves STRICTLY on the best so far (which, for the Ranger
    fine-tune run, starts at the first run's best) and counts the epochs since the last improvement."""

    def __init__(self, file, best, writer):
        self.file, self.best, self.writer = file, best, writer
        self.stale = 0

    def update(self, net, val_loss, bare_module):
        if val_loss < self.best:
            self.best, self.stale = val_loss, 0
            if self.writer:
                torch.save((net.module if bare_module else net).state_dict(), str(self.file))
            return True
        self.stale += 1
        return False

    def discard(self):
        if self.writer:
            try:
                os.remove(str(self.file))
            except FileNotFoundError:
                pass


# ---- data: who sees which crop -----------------------------------------------------------------------------------------------
class ShardPlan:
    """Batches of dataset indices for ONE rank and one phase.

    Single process: plain batches of ``batch_size`` (shuffled for 'train'), ``drop_last=False`` — the reference's
    DataLoader (train.py:365-371).  Under data parallelism ``batch_size`` is the global batch, as under ``nn.DataParallel``:
      * 'train': every rank needs the same number of steps (each step ends in a gradient all-reduce), so the shuffled
        index list is padded to a multiple of the world size (``DistributedSampler`` semantics) and dealt out in per-rank
        batches of ``batch_size // world``;
      * 'val': no crop is evaluated twice.  Global batch g = indices [g*B, (g+1)*B) is scattered like ``DataParallel``
        scatters a batch: contiguous chunks of ceil(m / world); a rank whose chunk is empty gets ``[]`` for that step
        and only joins the step's collectives."""

    def __init__(self, n, batch_size, world, rank, shuffle):
        self.n, self.world, self.rank, self.shuffle = n, world, rank, shuffle
        self.global_batch = max(int(batch_size), 1)
        self.local_batch = max(self.global_batch // world, 1)
        self.steps = None
        self.start_epoch(0)

    def start_epoch(self, epoch):
        """fixes this epoch's batches (``steps``: one index list per step, possibly empty under data parallelism)"""
        order = list(range(self.n))
        if self.shuffle and self.world == 1:
            order = torch.randperm(self.n).tolist()
        elif self.shuffle:                                # the same permutation on every rank, new every epoch
            order = torch.randperm(self.n, generator=torch.Generator().manual_seed(epoch)).tolist()
        if self.world == 1:
            self.steps = [order[i:i + self.global_batch] for i in range(0, self.n, self.global_batch)]
        elif self.shuffle:                                # train: padded, equal work on every rank
            total = math.ceil(self.n / self.world) * self.world
            order = (order * math.ceil(total / max(self.n, 1)))[:total]
            mine = order[self.rank::self.world]
            self.steps = [mine[i:i + self.local_batch] for i in range(0, len(mine), self.local_batch)]
        else:                                             # val: exact scatter of each global batch, nothing duplicated
            self.steps = []
            per_step = self.local_batch * self.world
            for lo in range(0, self.n, per_step):
                chunk = order[lo:lo + per_step]
                size = math.ceil(len(chunk) / self.world)
                self.steps.append(chunk[self.rank * size:(self.rank + 1) * size])

    def __iter__(self):                                   # the non-empty steps, as a DataLoader batch_sampler
        return iter([b for b in self.steps if b])

    def __len__(self):
        return sum(1 for b in self.steps if b)


def seed_worker(worker_id):
    """ Every DataLoader worker gets numpy / random seeds derived from torch's per-worker seed. """
    seed = torch.initial_seed() % 2 ** 32
    np.random.seed(seed)
    random.seed(seed)


class _Feeder:
    """One batch: pinned host tensors -> HBM, then (train phase only) the reference's Flip .. Noise + ToTensor pipeline
    on the device (training/device_augment.py)."""

    def __init__(self, label_type, device, train_transform):
        self.distance = label_type == 'distance'
        self.device = device
        self.augment = None
        if getattr(train_transform, 'device_augment', False):
            if device.type != 'cuda':
                raise RuntimeError("device augmentation needs the MI355X HIP path (no CPU fallback)")
            from .device_augment import DeviceAugment
            self.augment = DeviceAugment(label_type, train_transform.min_value, train_transform.max_value)

    def __call__(self, samples, training):
        """-> (image batch, tuple of label batches)"""
        img, *labels = (t.to(self.device, non_blocking=True) for t in samples)
        if not (training and self.augment is not None):
            return img, tuple(labels)
        if self.distance:
            img, labels = self.augment(img[:, 0], [(l[:, 0], 'linear') for l in labels])
            return img, tuple(l.unsqueeze(1) for l in labels)
        img, (label,) = self.augment(img[:, 0], [(labels[0], 'nearest')])
        return img, (label.to(torch.long),)


def _zip_trainset(path_data, zip_path):
    """train + val folders (not test) of the training set, deflated, next to the model"""
    root = path_data.stem
    with zipfile.ZipFile(zip_path, 'w', compression=zipfile.ZIP_DEFLATED) as archive:
        archive.write(path_data, arcname=root)
        for folder in sorted(p for p in path_data.iterdir() if p.stem != 'test'):
            archive.write(folder, arcname=os.path.join(root, folder.stem))
            for file in sorted(folder.glob('*')):
                archive.write(file, arcname=os.path.join(root, folder.stem, file.name))


def get_max_epochs(n_samples, crop_size):
    """ Maximum number of training epochs: a step table over the number of crops, made for 320 px crops and scaled by
    sqrt(320 / crop size), rounded down to a multiple of 20. """
    table = ((1000, 200), (500, 240), (200, 320), (100, 400), (50, 480), (0, 560))
    base = next(epochs for at_least, epochs in table if n_samples >= at_least)
    scaled = base * np.sqrt(320 / crop_size)
    return int(scaled - scaled % 20)


# ---- the worker -------------------------------------------------------------------------------------------------------------
class TrainWorker(QObject):
    """ Worker class for model training """
    finished = pyqtSignal()
    progress = pyqtSignal(int)
    text_output_main_gui = pyqtSignal(str)
    text_output = pyqtSignal(str)
    stop_training = False
    is_training = False
    num_workers = None      # None: 0 on CPU, min(cpu_count // 2, 16) otherwise (the reference's rule)
    precision = "fp32"      # "bf16": bf16 matrix-core operands, fp32 accumulate / statistics (BASELINE configs[2])
    graph_steps = False     # True: one GPU + Adam, the training step is recorded once and replayed (training/graph_step.py):
                            # 0.1-0.2 ms of host time per step instead of 7-18 ms.  Off by default: measured round 3, a step is
                            # GPU-bound even at batch 4 (bf16 8.0 ms eager with the weight gradients on a second stream,
                            # 8.7-8.8 ms replayed) — the replay buys host time (GUI thread, data loader), not throughput

    @pyqtSlot()
    def stop_training_process(self):
        """ Set internal training stop state to True """
        self.stop_training = True

    def _say(self, text, console):
        self.text_output.emit(text)
        if console:
            print(text)

    # -- one call per training request -------------------------------------------------------------------------------
    def start_training(self, path_data, path_models, label_type, iterations, optimizer, batch_size, device, num_gpus,
                       print_output=False, filters=None, max_epochs=None):
        """ Train ``iterations`` models on ``path_data/{train,val}`` and store them in ``path_models``. """
        from .. import engine
        with engine.precision_scope(self.precision):
            n_masks = {split: len(list((path_data / split).glob('mask*'))) for split in PHASES} \
                if any(path_data.glob('*')) else {'train': 0, 'val': 0}
            if min(n_masks.values()) >= 2:
                self.text_output_main_gui.emit('Start training')
                self.is_training = True
                for it in range(iterations):
                    QCoreApplication.processEvents()
                    if self.stop_training:
                        if self.is_training:
                            self.text_output_main_gui.emit("Stop training due to user interaction.")
                        break
                    run_name = unique_path(path_models, label_type + '_model_{:02d}.pth').stem
                    if label_type in ('boundary', 'distance'):
                        self._train_one_model(path_data, path_models, label_type, optimizer, run_name, device, num_gpus,
                                              MemoryLadder(batch_size, filters if filters is not None else [64, 1024]),
                                              (it, iterations), max_epochs, print_output)
                        batch_size = self._ladder_batch       # a reduced batch size sticks for the next iterations
                if not self.stop_training:
                    self.progress.emit(100)
            else:
                self.progress.emit(0)
        self.finished.emit()

    def _train_one_model(self, path_data, path_models, label_type, optimizer, run_name, device, num_gpus, ladder, where,
                         max_epochs, print_output):
        it, iterations = where
        distance = label_type == 'distance'
        while True:
            self._ladder_batch = ladder.batch_size
            configs = {'architecture': ('DU' if distance else 'U', "conv", 'mish' if optimizer == 'ranger' else 'relu',
                                        'bn', list(ladder.filters)),
                       'batch_size': ladder.batch_size, 'label_type': label_type,
                       'loss': 'smooth_l1' if distance else 'ce_dice', 'num_gpus': num_gpus, 'optimizer': optimizer,
                       'run_name': run_name}
            if self.precision != 'fp32':            # additive key: the reference's .json has none (it is fp32 only)
                configs['precision'] = self.precision
            if parallel.world_size() > 1:           # additive: how the global batch was dealt out
                configs['batch_size_per_gpu'] = max(ladder.batch_size // parallel.world_size(), 1)

            def fresh_net():
                unet_type, pool, act, norm, flt = configs['architecture']
                return build_unet(unet_type=unet_type, act_fun=act, pool_method=pool, normalization=norm, device=device,
                                  num_gpus=num_gpus, ch_in=1, ch_out=1 if distance else 3, filters=flt)
            try:
                transforms = augmentors(label_type=label_type, min_value=0, max_value=65535,
                                        device_augmentation=getattr(self, 'augment', True))
                configs['data_transforms'] = str(transforms)
                datasets = {x: TrainingDataset(root_dir=path_data, label_type=label_type, mode=x, transform=transforms[x])
                            for x in PHASES}
                if max_epochs is not None:
                    configs['max_epochs'] = int(max_epochs)
                else:
                    first_tif = next(iter((path_data / 'train').glob('*.tif')))
                    configs['max_epochs'] = get_max_epochs(len(datasets['train']) + len(datasets['val']),
                                                           crop_size=tiff.imread(str(first_tif)).shape[0])
                best = self.train(net=fresh_net(), datasets=datasets, configs=configs, device=device,
                                  path_models=path_models, train_progress=(1 / iterations, it / iterations),
                                  print_output=print_output)
                if optimizer == 'ranger' and self.is_training:
                    # second run: a fresh net + fresh optimizer start from the best weights, cosine annealing
                    parallel.barrier()              # rank 0 has written the checkpoint
                    tuned = get_weights(net=fresh_net(), weights=str(path_models / '{}.pth'.format(run_name)),
                                        num_gpus=num_gpus, device=device)
                    self.train(net=tuned, datasets=datasets, configs=configs, device=device, path_models=path_models,
                               best_loss=best, train_progress=(1 / iterations, (0.9 + it) / iterations),
                               print_output=print_output)
            except RuntimeError as err:
                if not is_out_of_memory(err):
                    raise
                out_of_memory = True
            else:
                out_of_memory = False
            if out_of_memory:
                # outside the handler: `err` (whose traceback pins the failed attempt's frames) is gone, the cyclic collector
                # frees the network / optimizer arenas / packed weights of that attempt, and only then is the cache emptied
                text = ladder.step_down()
                if ladder.exhausted:
                    if print_output:
                        print(text)
                    self.text_output_main_gui.emit(text)
                    self.text_output.emit('Stop training due to memory problems')
                self.text_output_main_gui.emit(text)
                datasets = None
                gc.collect()
                if torch.cuda.is_available():
                    torch.cuda.empty_cache()
                if ladder.exhausted:
                    return
                continue
            if self.is_training:                     # not stopped by the user: the run's side files
                self.progress.emit(int(100 * (it + 1) / iterations))
                if parallel.rank() == 0:
                    write_train_info(configs=configs, path=path_models)
                    _zip_trainset(path_data, path_models / '{}_trainset.zip'.format(run_name))
            return

    # -- one optimisation run ----------------------------------------------------------------------------------------------
    def _loaders(self, datasets, configs, device, world, rank):
        if self.num_workers is not None:
            workers = int(self.num_workers)
        elif device.type == "cpu":
            workers = 0
        else:
            try:
                workers = min(cpu_count() // 2, 16)
            except (AttributeError, NotImplementedError):
                workers = 4
        plans = {x: ShardPlan(len(datasets[x]), configs['batch_size'], world, rank, shuffle=(x == 'train')) for x in PHASES}
        loaders = {x: torch.utils.data.DataLoader(datasets[x], batch_sampler=plans[x], pin_memory=True,
                                                  worker_init_fn=seed_worker, num_workers=workers) for x in PHASES}
        return plans, loaders

    def _run_phase(self, phase, net, plan, loader, feeder, criterion, regime, distance, device, world, n_total):
        """all batches of one phase; returns the mean loss per crop over the WHOLE (global) phase"""
        training = phase == 'train'
        net.train() if training else net.eval()
        if world > 1 and not training:
            parallel.sync_eval_buffers(net)          # validate the model rank 0 would save (all ranks, even idle ones)
        loss_sum, seen = 0.0, 0
        loss = None
        batches = iter(loader)
        graphed = regime.graphed_step(net, criterion, distance, device, world) if training and self.graph_steps else None
        for indices in plan.steps:
            if not indices:                          # validation step in which this rank's scatter chunk is empty
                if not distance:
                    parallel.allreduce_dice_sums(torch.zeros(6, dtype=torch.float64, device=device), 0.0)
                continue
            img, labels = feeder(next(batches), training)
            if graphed is not None:
                loss = graphed(img, *labels)
                loss_sum += float(loss.item() * img.size(0))
                seen += img.size(0)
                continue
            regime.optimizer.zero_grad()
            with torch.set_grad_enabled(training):
                if distance:
                    border, cell = net(img)
                    loss = criterion['border'](border, labels[0]) + criterion['cell'](cell, labels[1])
                else:
                    loss = criterion(net(img), labels[0])
                if training:
                    loss.backward()
                    regime.optimizer.step()
            loss_sum += float(loss.item() * img.size(0))       # one device -> host sync per step, as in the reference
            seen += img.size(0)
        if world > 1:
            loss_sum = parallel.allreduce_scalar_sum(loss_sum, device)
            seen = parallel.allreduce_scalar_sum(float(seen), device)
            return loss_sum / max(seen, 1.0)
        return loss_sum / n_total

    def train(self, net, datasets, configs, device, path_models, train_progress, best_loss=1e4, print_output=False):
        """ Train the model; returns the best validation loss.  ``best_loss < 1e3`` marks the Ranger fine-tune run. """
        device = torch.device(device)
        world, rank = parallel.world_size(), parallel.rank()
        console = print_output and rank == 0
        finetune = best_loss < 1e3
        distance = configs['label_type'] == 'distance'
        run = configs['run_name']

        if finetune:
            self.text_output.emit('Start 2nd run with cosine annealing')
            if console:
                print('   Start 2nd run with cosine annealing')
        else:
            sizes = 'Train/validate on {}/{} images'.format(len(datasets['train']), len(datasets['val']))
            for gui_line, console_line in (('-' * 10, '*' * 10), (run, run), ('-' * 10, '*' * 10), (sizes, sizes)):
                self.text_output.emit('{}'.format(gui_line))
                if console:
                    print('{}'.format(console_line))

        feeder = _Feeder(configs['label_type'], device, getattr(datasets['train'], 'transform', None))
        plans, loaders = self._loaders(datasets, configs, device, world, rank)
        criterion = get_loss(configs['loss'], label_type=configs['label_type'])
        regime = _Regime(configs['optimizer'], finetune, net.parameters(), configs['max_epochs'])
        keeper = BestCheckpoint(path_models / (run + '.pth'), best_loss, writer=rank == 0)
        history = []                                  # (train loss, val loss) per finished epoch
        started = time.time()

        for epoch in range(regime.epochs):
            QCoreApplication.processEvents()
            if self.stop_training:
                self.text_output_main_gui.emit("Stop training due to user interaction.\nRemove last model.")
                self.text_output.emit("Stop training due to user interaction.")
                keeper.discard()
                self.is_training = False
                break
            plans['train'].start_epoch(epoch)
            means = {phase: self._run_phase(phase, net, plans[phase], loaders[phase], feeder, criterion, regime, distance,
                                            device, world, len(datasets[phase])) for phase in PHASES}
            history.append((means['train'], means['val']))
            line = '{} / {}: Loss train / val: {:.4f} / {:.4f}'.format(epoch + 1, regime.epochs, *history[-1])
            if keeper.update(net, means['val'], bare_module=configs['num_gpus'] > 1):
                line += ' --> save'
            self._say(line, console)
            regime.after_validation(means['val'])
            self.progress.emit(regime.percent(epoch + 1, train_progress))
            if keeper.stale == regime.stale_limit:
                self._say('{} epochs without val loss improvement --> break'.format(keeper.stale), console)
                break

        if not self.stop_training:
            elapsed = time.time() - started
            self._say('Training completed in {:.0f}min {:.0f}s'.format(elapsed // 60, elapsed % 60), console)
            table = np.array([(i + 1, tr, va) for i, (tr, va) in enumerate(history)], dtype=np.float64).reshape(-1, 3)
            log = path_models / (run + '_loss.txt')
            keys = ('training_time_run_2', 'trained_epochs_run2') if finetune else ('training_time', 'trained_epochs')
            if rank == 0:
                if finetune:                          # appended below the first run's table
                    with open(str(log), 'a') as f:
                        f.write('\n')
                        np.savetxt(f, X=table, fmt=['%3i', '%2.5f', '%2.5f'], delimiter=',')
                else:
                    np.savetxt(fname=str(log), X=table, fmt=['%3i', '%2.5f', '%2.5f'],
                               header='Epoch, training loss, validation loss', delimiter=',')
            configs[keys[0]], configs[keys[1]] = elapsed, len(history)

        del net, regime, loaders
        gc.collect()
        return keeper.best

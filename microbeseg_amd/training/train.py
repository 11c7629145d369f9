"""Training driver of the hot path: optimizer / scheduler set-up, epoch / phase / batch loop, best-validation
checkpointing, early stopping, loss log, OOM back-off ladder, second (cosine) Ranger run.

Mirror of ``TrainWorker`` in ``src/training/train.py`` (reference: ``start_training`` :124-304, ``train`` :316-576,
``get_max_epochs`` :579-606, ``seed_worker`` :609-620).  Same public names, arguments, signals, messages and files
(``<run>.pth`` state dict of the bare module, ``<run>.json``, ``<run>_loss.txt``, ``<run>_trainset.zip``;
SURVEY.md Appendix D).  Differences, all additive:
  * Qt is optional (``utils.qt_shim``) so the worker runs headless;
  * ``start_training(..., filters=None, max_epochs=None)`` lets callers pick the network size / epoch budget
    (the reference hard-codes ``[64, 1024]``; BASELINE configs[0] needs a 2-level, 8-filter net);
  * with ``num_gpus > 1`` under torch.distributed each rank trains on its shard of the crop list, validation losses
    are summed over ranks so every rank takes identical save / early-stop decisions, and rank 0 writes the files;
  * ``CreateLabelsWorker`` (reference :26-112) creates the labels on the device (train_data_representations.py);
  * the crop batch is augmented on the device in the train phase (device_augment.py).
"""
import gc
import os
import random
import time
import zipfile
from multiprocessing import cpu_count

import numpy as np
import torch
import torch.optim as optim
from torch.optim.lr_scheduler import ReduceLROnPlateau, CosineAnnealingLR

from ..utils import tiffio as tiff
from ..utils.qt_shim import QObject, QCoreApplication, pyqtSignal, pyqtSlot
from ..utils.unets import build_unet, get_weights
from ..utils.utils import unique_path, write_train_info
from .. import parallel
from .losses import get_loss
from .ranger2020 import Ranger
from .training_dataset import TrainingDataset, augmentors
from .train_data_representations import get_label, max_major_axis_length
from shutil import rmtree


class CreateLabelsWorker(QObject):
    """ Worker class for label creation (reference src/training/train.py:26-112): every ``mask*.tif`` of
    ``path/{train,val}`` gets its training representation written next to it (``cell_dist_*`` + ``neighbor_dist_*`` for
    the distance method, ``<label_type>_*`` otherwise). """
    finished = pyqtSignal()
    progress = pyqtSignal(int)
    text_output = pyqtSignal(str)
    stop_label_creation = False

    def create_labels(self, path, label_type):
        if len(list(path.glob('*'))) == 0:            # export has been stopped (folders are deleted)
            self.progress.emit(0)
            self.finished.emit()
            return
        self.text_output.emit('Create labels')
        mask_ids_train = list((path / 'train').glob('mask*.tif'))
        mask_ids_val = list((path / 'val').glob('mask*.tif'))
        if len(mask_ids_val) < 2 or len(mask_ids_train) < 2:
            self.text_output.emit("The training and the validation set should each contain at least two annotated "
                                  "images! Stop")
            self.progress.emit(0)
            self.finished.emit()
            return
        mask_ids = mask_ids_train + mask_ids_val
        for i, mask_id in enumerate(mask_ids):
            QCoreApplication.processEvents()          # update to get the stop signal
            if self.stop_label_creation:
                self.text_output.emit("Stop label creation due to user interaction.\nDelete local folder.")
                rmtree(str(path))
                break
            mask = tiff.imread(str(mask_id))
            if label_type == 'distance':              # search radius from the largest major axis length
                max_mal = int(np.ceil(max_major_axis_length(mask)))
            else:
                max_mal = 0
            label = get_label(mask=mask, label_type=label_type, max_mal=max_mal)
            fname = mask_id.name.split('mask_')[-1]
            if label_type == 'distance':
                tiff.imwrite(str(mask_id.parent / 'cell_dist_{}'.format(fname)), label[0])
                tiff.imwrite(str(mask_id.parent / 'neighbor_dist_{}'.format(fname)), label[1])
            else:
                tiff.imwrite(str(mask_id.parent / '{}_{}'.format(label_type, fname)), label)
            self.progress.emit(int(100 * (i + 1) / len(mask_ids)))
        if self.stop_label_creation:
            self.progress.emit(0)
        else:
            self.progress.emit(100)
        self.finished.emit()
        return

    @pyqtSlot()
    def stop_label_creation_process(self):
        self.stop_label_creation = True


class TrainWorker(QObject):
    """ Worker class for model training """
    finished = pyqtSignal()
    progress = pyqtSignal(int)
    text_output_main_gui = pyqtSignal(str)
    text_output = pyqtSignal(str)
    stop_training = False
    is_training = False
    num_workers = None      # None -> reference rule (0 on CPU, min(cpu_count // 2, 16) otherwise)
    precision = "fp32"      # "bf16": bf16 matrix-core inputs, fp32 accumulate / storage / statistics (BASELINE configs[2])

    def start_training(self, path_data, path_models, label_type, iterations, optimizer, batch_size, device, num_gpus,
                       print_output=False, filters=None, max_epochs=None):
        """ Train ``iterations`` models on ``path_data/{train,val}`` and store them in ``path_models``. """
        from .. import engine
        previous = engine.get_precision()
        engine.set_precision(self.precision)
        try:
            return self._start_training(path_data, path_models, label_type, iterations, optimizer, batch_size, device,
                                        num_gpus, print_output, filters, max_epochs)
        finally:
            engine.set_precision(previous)

    def _start_training(self, path_data, path_models, label_type, iterations, optimizer, batch_size, device, num_gpus,
                        print_output=False, filters=None, max_epochs=None):
        if len(list(path_data.glob('*'))) == 0 or len(list((path_data / 'train').glob('mask*'))) < 2 \
                or len(list((path_data / 'val').glob('mask*'))) < 2:
            self.progress.emit(0)
            self.finished.emit()
            return

        self.text_output_main_gui.emit('Start training')
        self.is_training = True
        rank0 = parallel.rank() == 0

        for i in range(iterations):
            QCoreApplication.processEvents()
            if self.stop_training:
                if self.is_training:
                    self.text_output_main_gui.emit("Stop training due to user interaction.")
                break

            run_name = unique_path(path_models, label_type + '_model_{:02d}.pth').stem
            act_fun = 'mish' if optimizer == 'ranger' else 'relu'
            if label_type not in ['boundary', 'distance']:
                continue

            net_filters = list(filters) if filters is not None else [64, 1024]
            try_training = True
            while try_training:
                try:
                    train_configs = {'architecture': ('DU' if label_type == 'distance' else 'U', "conv", act_fun, 'bn',
                                                      net_filters),
                                     'batch_size': batch_size,
                                     'label_type': label_type,
                                     'loss': 'smooth_l1' if label_type == 'distance' else 'ce_dice',
                                     'num_gpus': num_gpus,
                                     'optimizer': optimizer,
                                     'run_name': run_name}
                    if self.precision != 'fp32':          # additive key; the reference's .json has none (fp32 only)
                        train_configs['precision'] = self.precision

                    def make_net():
                        return build_unet(unet_type=train_configs['architecture'][0],
                                          act_fun=train_configs['architecture'][2],
                                          pool_method=train_configs['architecture'][1],
                                          normalization=train_configs['architecture'][3],
                                          device=device, num_gpus=num_gpus, ch_in=1,
                                          ch_out=1 if label_type == 'distance' else 3,
                                          filters=train_configs['architecture'][4])

                    net = make_net()
                    data_transforms = augmentors(label_type=label_type, min_value=0, max_value=65535,
                                                 device_augmentation=getattr(self, 'augment', True))
                    train_configs['data_transforms'] = str(data_transforms)
                    datasets = {x: TrainingDataset(root_dir=path_data, label_type=label_type, mode=x,
                                                   transform=data_transforms[x]) for x in ['train', 'val']}
                    crop_size = tiff.imread(str(list((path_data / 'train').glob('*.tif'))[0])).shape[0]
                    train_configs['max_epochs'] = int(max_epochs) if max_epochs is not None else \
                        get_max_epochs(len(datasets['train']) + len(datasets['val']), crop_size=crop_size)

                    best_loss = self.train(net=net, datasets=datasets, configs=train_configs, device=device,
                                           path_models=path_models, train_progress=(1 / iterations, i / iterations),
                                           print_output=print_output)

                    if train_configs['optimizer'] == 'ranger' and self.is_training:
                        # fine-tune a fresh net that starts from the best weights, cosine annealing (train.py:229-252)
                        net = make_net()
                        parallel.barrier()
                        net = get_weights(net=net, weights=str(path_models / '{}.pth'.format(run_name)),
                                          num_gpus=num_gpus, device=device)
                        _ = self.train(net=net, datasets=datasets, configs=train_configs, device=device,
                                       path_models=path_models, best_loss=best_loss,
                                       train_progress=(1 / iterations, (0.9 + i) / iterations),
                                       print_output=print_output)
                    try_training = False

                    if self.is_training:
                        self.progress.emit(int(100 * (i + 1) / iterations))
                        if rank0:
                            write_train_info(configs=train_configs, path=path_models)
                            _zip_trainset(path_data, path_models / '{}_trainset.zip'.format(run_name))

                except RuntimeError as e:  # out of memory -> back-off ladder (train.py:276-297)
                    if not _is_oom(e):
                        raise
                    if batch_size > 8:
                        text = "Model does not fit on RAM/VRAM. Reduce batch size from {} to 8".format(batch_size)
                        batch_size = 8
                    elif batch_size > 4:
                        text = "Model does not fit on RAM/VRAM. Reduce batch size from {} to 4".format(batch_size)
                        batch_size = 4
                    elif net_filters[0] > 32:
                        text = "Model does not fit on RAM/VRAM. Reduce number of kernels"
                        net_filters = [32, 512]
                    elif net_filters[-1] == 512:
                        text = "Model does not fit on RAM/VRAM. Reduce model depth"
                        net_filters = [32, 256]
                    else:
                        text = "Please, try again with smaller batch size or reduce the crop size (use the export " \
                               "and import functionalities for this)"
                        if print_output:
                            print(text)
                        self.text_output_main_gui.emit(text)
                        self.text_output.emit('Stop training due to memory problems')
                        try_training = False
                    self.text_output_main_gui.emit(text)
                    torch.cuda.empty_cache() if torch.cuda.is_available() else None

        if not self.stop_training:
            self.progress.emit(100)
        self.finished.emit()
        return

    @pyqtSlot()
    def stop_training_process(self):
        """ Set internal training stop state to True """
        self.stop_training = True

    def train(self, net, datasets, configs, device, path_models, train_progress, best_loss=1e4, print_output=False):
        """ Train the model; returns the best validation loss. """
        device = torch.device(device)
        world, rank = parallel.world_size(), parallel.rank()
        rank0 = rank == 0
        print_output = print_output and rank0
        # 'train' transform of the reference = augmentation + ToTensor (mytransforms.py:24-31); here the DataLoader delivers
        # raw crops and the augmentation runs on the device, batch-wise (training/device_augment.py)
        device_augment = None
        train_tf = getattr(datasets['train'], 'transform', None)
        if getattr(train_tf, 'device_augment', False):
            if device.type != 'cuda':
                raise RuntimeError("device augmentation needs the MI355X HIP path (no CPU fallback)")
            from .device_augment import DeviceAugment
            device_augment = DeviceAugment(configs['label_type'], train_tf.min_value, train_tf.max_value)

        if best_loss < 1e3:  # second Ranger run
            second_run = True
            self.text_output.emit('Start 2nd run with cosine annealing')
            if print_output:
                print('   Start 2nd run with cosine annealing')
        else:
            second_run = False
            self.text_output.emit('-' * 10)
            self.text_output.emit('{}'.format(configs['run_name']))
            self.text_output.emit('-' * 10)
            self.text_output.emit('Train/validate on {}/{} images'.format(len(datasets['train']), len(datasets['val'])))
            if print_output:
                print('*' * 10)
                print('{}'.format(configs['run_name']))
                print('*' * 10)
                print('Train/validate on {}/{} images'.format(len(datasets['train']), len(datasets['val'])))

        if self.num_workers is not None:
            num_workers = int(self.num_workers)
        elif device.type == "cpu":
            num_workers = 0
        else:
            try:
                num_workers = cpu_count() // 2
            except (AttributeError, NotImplementedError):
                num_workers = 4
            num_workers = int(np.minimum(num_workers, 16))
        samplers = {x: None for x in ['train', 'val']}
        if world > 1:
            from torch.utils.data.distributed import DistributedSampler
            samplers = {'train': DistributedSampler(datasets['train'], world, rank, shuffle=True, drop_last=False),
                        'val': DistributedSampler(datasets['val'], world, rank, shuffle=False, drop_last=False)}
        dataloader = {x: torch.utils.data.DataLoader(datasets[x], batch_size=configs['batch_size'],
                                                     shuffle=(x == 'train' and samplers[x] is None),
                                                     sampler=samplers[x], pin_memory=True,
                                                     worker_init_fn=seed_worker, num_workers=num_workers)
                      for x in ['train', 'val']}

        criterion = get_loss(configs['loss'], label_type=configs['label_type'])
        max_epochs = configs['max_epochs']

        if configs['optimizer'] == 'adam':
            optimizer = optim.Adam(net.parameters(), lr=8e-4, betas=(0.9, 0.999), eps=1e-08, weight_decay=0,
                                   amsgrad=True)
            scheduler = ReduceLROnPlateau(optimizer, mode='min', factor=0.25, patience=configs['max_epochs'] // 20,
                                          min_lr=3e-6)
            break_condition = 2 * configs['max_epochs'] // 20 + 5
        elif configs['optimizer'] == 'ranger':
            lr = 6e-3
            if second_run:
                optimizer = Ranger(net.parameters(), lr=0.09 * lr, alpha=0.5, k=6, N_sma_threshhold=5,
                                   betas=(.95, 0.999), eps=1e-6, weight_decay=0, use_gc=True, gc_conv_only=False,
                                   gc_loc=True)
                scheduler = CosineAnnealingLR(optimizer, T_max=max(1, configs['max_epochs'] // 10), eta_min=3e-5,
                                              last_epoch=-1)
                break_condition = configs['max_epochs'] // 10 + 1
                max_epochs = configs['max_epochs'] // 10
            else:
                optimizer = Ranger(net.parameters(), lr=lr, alpha=0.5, k=6, N_sma_threshhold=5, betas=(.95, 0.999),
                                   eps=1e-6, weight_decay=0, use_gc=True, gc_conv_only=False, gc_loc=True)
                scheduler = ReduceLROnPlateau(optimizer, mode='min', factor=0.25,
                                              patience=configs['max_epochs'] // 10, min_lr=0.075 * lr)
                break_condition = 2 * configs['max_epochs'] // 10 + 5
        else:
            raise Exception('Optimizer not known')

        epochs_wo_improvement, train_loss, val_loss = 0, [], []
        since = time.time()
        epoch = -1
        loss = None

        for epoch in range(max_epochs):
            QCoreApplication.processEvents()
            if self.stop_training:
                self.text_output_main_gui.emit("Stop training due to user interaction.\nRemove last model.")
                self.text_output.emit("Stop training due to user interaction.")
                if rank0:
                    try:
                        os.remove(str(path_models / "{}.pth".format(configs['run_name'])))
                    except FileNotFoundError:
                        pass
                self.is_training = False
                break

            for phase in ['train', 'val']:
                net.train() if phase == 'train' else net.eval()
                if samplers[phase] is not None and phase == 'train':
                    samplers[phase].set_epoch(epoch)
                running_loss = 0.0
                seen = 0

                dev_aug = device_augment if phase == 'train' else None
                for samples in dataloader[phase]:
                    if configs['label_type'] == 'distance':
                        img_batch, border_label_batch, cell_label_batch = samples
                        img_batch = img_batch.to(device, non_blocking=True)
                        cell_label_batch = cell_label_batch.to(device, non_blocking=True)
                        border_label_batch = border_label_batch.to(device, non_blocking=True)
                        if dev_aug is not None:      # Flip .. Noise + ToTensor of the reference's 'train' transform, on the device
                            img_batch, (border_label_batch, cell_label_batch) = dev_aug(
                                img_batch[:, 0], [(border_label_batch[:, 0], 'linear'), (cell_label_batch[:, 0], 'linear')])
                            border_label_batch = border_label_batch.unsqueeze(1)
                            cell_label_batch = cell_label_batch.unsqueeze(1)
                    else:
                        img_batch, label_batch = samples
                        img_batch = img_batch.to(device, non_blocking=True)
                        label_batch = label_batch.to(device, non_blocking=True)
                        if dev_aug is not None:
                            img_batch, (label_batch,) = dev_aug(img_batch[:, 0], [(label_batch, 'nearest')])
                            label_batch = label_batch.to(torch.long)

                    optimizer.zero_grad()
                    with torch.set_grad_enabled(phase == 'train'):
                        if configs['label_type'] == 'distance':
                            border_pred_batch, cell_pred_batch = net(img_batch)
                            loss_border = criterion['border'](border_pred_batch, border_label_batch)
                            loss_cell = criterion['cell'](cell_pred_batch, cell_label_batch)
                            loss = loss_border + loss_cell
                        else:
                            pred_batch = net(img_batch)
                            loss = criterion(pred_batch, label_batch)
                        if phase == 'train':
                            loss.backward()
                            optimizer.step()
                    running_loss += float(loss.item() * img_batch.size(0))
                    seen += img_batch.size(0)

                if world > 1:   # global epoch loss -> identical decisions on every rank
                    running_loss = parallel.allreduce_scalar_sum(running_loss, device)
                    seen = parallel.allreduce_scalar_sum(float(seen), device)
                    epoch_loss = running_loss / max(seen, 1.0)
                else:
                    epoch_loss = running_loss / len(datasets[phase])

                if phase == 'train':
                    train_loss.append(epoch_loss)
                else:
                    val_loss.append(epoch_loss)
                    if epoch_loss < best_loss:
                        train_output = '{} / {}: Loss train / val: {:.4f} / {:.4f} --> save'.format(
                            epoch + 1, max_epochs, train_loss[-1], epoch_loss)
                        best_loss = epoch_loss
                        if rank0:   # always the state dict of the bare module (train.py:512-515)
                            bare = net.module if configs['num_gpus'] > 1 else net
                            torch.save(bare.state_dict(), str(path_models / (configs['run_name'] + '.pth')))
                        epochs_wo_improvement = 0
                    else:
                        train_output = '{} / {}: Loss train / val: {:.4f} / {:.4f}'.format(
                            epoch + 1, max_epochs, train_loss[-1], epoch_loss)
                        epochs_wo_improvement += 1
                    self.text_output.emit(train_output)
                    if print_output:
                        print(train_output)
                    if configs['optimizer'] == 'ranger' and second_run:
                        scheduler.step()
                    else:
                        scheduler.step(epoch_loss)

            if configs['optimizer'] == 'ranger':
                if not second_run:
                    self.progress.emit(
                        int(100 * (epoch + 1) / (1.1 * max_epochs) * train_progress[0] + 100 * train_progress[1]))
                else:
                    self.progress.emit(
                        int(100 * (epoch + 1) / (10 * max_epochs) * train_progress[0] + 100 * train_progress[1]))
            else:
                self.progress.emit(int(100 * (epoch + 1) / max_epochs * train_progress[0] + 100 * train_progress[1]))

            if epochs_wo_improvement == break_condition:
                self.text_output.emit('{} epochs without val loss improvement --> break'.format(epochs_wo_improvement))
                if print_output:
                    print('{} epochs without val loss improvement --> break'.format(epochs_wo_improvement))
                break

        if not self.stop_training:
            time_elapsed = time.time() - since
            self.text_output.emit('Training completed in {:.0f}min {:.0f}s'.format(time_elapsed // 60, time_elapsed % 60))
            if print_output:
                print('Training completed in {:.0f}min {:.0f}s'.format(time_elapsed // 60, time_elapsed % 60))
            stats = np.transpose(np.array([list(range(1, len(train_loss) + 1)), train_loss, val_loss]))
            if second_run:
                if rank0:
                    with open(str(path_models / (configs['run_name'] + '_loss.txt')), 'a') as f:
                        f.write('\n')
                        np.savetxt(f, X=stats, fmt=['%3i', '%2.5f', '%2.5f'], delimiter=',')
                configs['training_time_run_2'], configs['trained_epochs_run2'] = time_elapsed, epoch + 1
            else:
                if rank0:
                    np.savetxt(fname=str(path_models / (configs['run_name'] + '_loss.txt')), X=stats,
                               fmt=['%3i', '%2.5f', '%2.5f'], header='Epoch, training loss, validation loss',
                               delimiter=',')
                configs['training_time'], configs['trained_epochs'] = time_elapsed, epoch + 1

        del net, loss, optimizer, scheduler
        gc.collect()
        return best_loss


def _is_oom(err):
    """The reference treats every RuntimeError as 'does not fit'; keep its ladder for real memory errors only so that
    genuine bugs (bad shapes, missing library) surface instead of silently shrinking the model."""
    msg = str(err).lower()
    return "out of memory" in msg or "hiperroroutofmemory" in msg or "hip error: out of memory" in msg


def _zip_trainset(path_data, zip_path):
    with zipfile.ZipFile(zip_path, 'w') as z:
        z.write(path_data, arcname=path_data.stem, compress_type=zipfile.ZIP_DEFLATED)
        for sub_dir in path_data.iterdir():
            if sub_dir.stem == 'test':
                continue
            z.write(sub_dir, arcname=os.path.join(path_data.stem, sub_dir.stem), compress_type=zipfile.ZIP_DEFLATED)
            for file in sub_dir.glob('*'):
                z.write(file, arcname=os.path.join(path_data.stem, sub_dir.stem, file.name),
                        compress_type=zipfile.ZIP_DEFLATED)


def get_max_epochs(n_samples, crop_size):
    """ Maximum number of training epochs (heuristic made for 320x320 px crops; reference train.py:579-606). """
    for bound, epochs in ((1000, 200), (500, 240), (200, 320), (100, 400), (50, 480)):
        if n_samples >= bound:
            max_epochs = epochs
            break
    else:
        max_epochs = 560
    max_epochs *= np.sqrt(320 / crop_size)
    return int(max_epochs - max_epochs % 20)


def seed_worker(worker_id):
    """ Give every DataLoader worker its own numpy / random seed (reference train.py:609-620). """
    worker_seed = torch.initial_seed() % 2 ** 32
    np.random.seed(worker_seed)
    random.seed(worker_seed)

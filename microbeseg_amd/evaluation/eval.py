"""Model evaluation on the test split: batched inference, threshold sweep of the post-processing, AJI+ scoring.

Drop-in for ``EvalWorker`` of the reference (``src/evaluation/eval.py``: ``start_evaluation`` :36-196, ``calc_scores``
:222-322, ``inference`` :324-425): same method names, arguments, signals, messages and files —
``<results>/<set>_<model>/{mask*.tif, [raw*.tif], scores.csv, test_set.zip}`` and ``<results>.csv`` with the columns
``model, th_cell, th_seed, aji+ (mean), aji+ (std), test set version`` — and the same selection rule: for the distance
method the threshold pair of the 4 x 2 grid with the best mean AJI+ is kept.  Written from that behaviour (SURVEY.md §3.3,
Appendix D); the structure is this build's own:

  ``THRESHOLD_GRID``   the grid as data;       ``_ModelUnderTest``  checkpoint + json -> network in HBM
  ``_ScoreTable``      rows of ``<results>.csv`` incl. the merge with an earlier file of the same test-set version
  ``EvalWorker``       the worker: ``inference`` writes masks, ``calc_scores`` scores and promotes the best directory

MI355X path (the direct caller of the hot path): the forward of a batch and the post-processing of every
image stay in HBM; the 8 threshold pairs of an image are ONE call (``mseg_distance_postprocess_sweep``: smoothing shared),
only uint16 masks return; scoring = border correction + relabelling + area / intersection counts on the device
(``mseg_eval_relabel``, ``mseg_eval_pair_counts``), the Hungarian pairing on the small IoU matrix stays scipy on the host
as in the reference.  ``num_gpus`` is accepted for signature compatibility; a worker evaluates on its own GPU.
"""
import gc
import hashlib
import json
import os
import shutil
import zipfile
from itertools import product
from multiprocessing import cpu_count

import numpy as np
import pandas as pd
import torch

from .. import _lib
from ..inference import postprocessing as pp
from ..inference.inference_dataset import InferenceDataset, pre_processing_transforms
from ..utils import tiffio as tiff
from ..utils.qt_shim import QCoreApplication, QObject, pyqtSignal, pyqtSlot
from ..utils.unets import build_unet, get_weights
from .stats_utils import aji_plus_masks

# (th_cell, th_seed) pairs tried for a distance model; a boundary model has no thresholds (one pseudo entry, -1)
THRESHOLD_GRID = {'distance': list(product([0.05, 0.075, 0.10, 0.125], [0.35, 0.45])), 'boundary': [-1]}
MEMORY_HINT = ("Please, try again with smaller batch size or reduce the crop size (use the export and import "
               "functionalities for this)")
CSV_COLUMNS = ('model', 'th_cell', 'th_seed', 'aji+ (mean)', 'aji+ (std)', 'test set version')


class _ModelUnderTest:
    """``<dir>/<name>.pth`` + ``<name>.json`` -> (network in eval mode on ``device``, label type)"""

    def __init__(self, checkpoint, device):
        with open(checkpoint.parent / "{}.json".format(checkpoint.stem)) as f:
            settings = json.load(f)
        unet_type, pool, act, norm, filters = settings['architecture'][:5]
        self.label_type = settings['label_type']
        net = build_unet(unet_type=unet_type, act_fun=act, pool_method=pool, normalization=norm, device=device,
                         num_gpus=1, ch_in=1, ch_out=1 if self.label_type == 'distance' else 3, filters=filters)
        self.net = get_weights(net=net, weights=str(checkpoint), num_gpus=1, device=device)
        self.title = "{}: {}".format(checkpoint.parent.stem, checkpoint.stem)


class _ScoreTable:
    """one row per evaluated model; ``write`` merges with the rows an earlier evaluation left for the SAME test-set
    version (rows of other versions are dropped, a re-evaluated model replaces its old row) and sorts by model"""

    def __init__(self):
        self.rows = []

    def add(self, title, mean, std, th_cell, th_seed, version):
        self.rows.append(dict(zip(CSV_COLUMNS, (title, th_cell, th_seed, mean, std, version))))

    def write(self, csv_path):
        table = pd.DataFrame(self.rows, columns=list(CSV_COLUMNS))
        if csv_path.is_file():
            earlier = pd.read_csv(csv_path)
            same_version = earlier[earlier['test set version'] == table.iloc[0]['test set version']]
            table = pd.concat([table, same_version]).drop_duplicates('model')
        table.sort_values(by=['model']).to_csv(csv_path, header=True, index=False)


def _archive_test_set(path_data, archive_path):
    root = path_data.stem
    with zipfile.ZipFile(archive_path, 'w', compression=zipfile.ZIP_DEFLATED) as archive:
        archive.write(path_data, arcname=root)
        archive.write(path_data / 'test', arcname=os.path.join(root, 'test'))
        for file in sorted((path_data / 'test').glob('*')):
            archive.write(file, arcname=os.path.join(root, 'test', file.name))


class EvalWorker(QObject):
    """ Worker class for model evaluation """
    finished = pyqtSignal()
    progress = pyqtSignal(int)
    text_output = pyqtSignal(str)
    stop_evaluation = False
    is_evaluating = False

    @pyqtSlot()
    def stop_evaluation_process(self):
        """ Set internal evaluation stop state to True """
        self.stop_evaluation = True

    def _leave(self, percent=None):
        if percent is not None:
            self.progress.emit(percent)
        self.finished.emit()

    def start_evaluation(self, path_data, path_results, models, batch_size, device, num_gpus, save_raw_pred,
                         start_message=''):
        """ Evaluate ``models`` (paths of ``.pth`` checkpoints next to their ``.json``) on ``path_data / 'test'``. """
        if not path_data.is_dir():
            return self._leave(0)
        if len(list((path_data / 'test').glob('mask*'))) < 2:
            self.text_output.emit('Not enough test images found. At least 2 are needed (better more)')
            return self._leave(0)
        self.is_evaluating = True
        self.text_output.emit(start_message)
        table = _ScoreTable()
        share = 1 / len(models)
        for done, checkpoint in enumerate(models):
            out_dir = path_results / "{}_{}".format(checkpoint.parent.stem, checkpoint.stem)
            path_results.mkdir(exist_ok=True)
            if out_dir.is_dir():
                shutil.rmtree(out_dir)
            out_dir.mkdir()
            QCoreApplication.processEvents()
            if self.stop_evaluation:
                if self.is_evaluating:
                    self.text_output.emit("Stop evaluation due to user interaction.")
                break
            model = _ModelUnderTest(checkpoint, device)
            test_set = InferenceDataset(data_dir=path_data / 'test',
                                        transform=pre_processing_transforms(apply_clahe=False, scale_factor=1))
            try:
                self.inference(net=model.net, dataset=test_set, label_type=model.label_type,
                               ths=THRESHOLD_GRID.get(model.label_type, [-1]), batch_size=batch_size, device=device,
                               path_model=out_dir, save_raw=save_raw_pred, eval_progress=(0.5 * share, done * share))
            except RuntimeError as err:
                if not _is_out_of_memory(err):
                    raise
                for line in (MEMORY_HINT, 'Stop evaluation due to memory problems', MEMORY_HINT):
                    self.text_output.emit(line)
                return self._leave()
            label_type = model.label_type
            title = model.title
            del model
            gc.collect()
            outcome = self.calc_scores(prediction_path=out_dir, test_set_path=path_data / 'test', label_type=label_type)
            if outcome:
                table.add(title, *outcome)
                _archive_test_set(path_data, out_dir / 'test_set.zip')
            self.progress.emit(int(100 * (done + 1) * share))
        if table.rows and not self.stop_evaluation:
            table.write(path_results.parent / '{}.csv'.format(path_results.stem))
            self.progress.emit(100)
        self.is_evaluating = False
        self._leave()

    # -- scoring -----------------------------------------------------------------------------------------------------------
    def _aji_of_directory(self, directory, test_set_path):
        """{mask stem: AJI+} for every ``mask*.tif`` of ``directory`` (None when the user stopped the evaluation)"""
        scores = {}
        for mask_file in directory.glob('mask*.tif'):
            QCoreApplication.processEvents()
            if self.stop_evaluation:
                if self.is_evaluating:
                    self.text_output.emit("Stop metric calculation.")
                return None
            # border correction, connected-component relabelling and the AJI+ statistics run fused on the device
            scores[mask_file.stem] = aji_plus_masks(tiff.imread(str(mask_file)),
                                                    tiff.imread(str(test_set_path / mask_file.name)))
        return scores

    def calc_scores(self, prediction_path, test_set_path, label_type):
        """ AJI+ per test image.  Distance method: every ``<th_cell>_<th_seed>`` sub-directory is scored, the one with the
        best mean is kept (its files move up, all sub-directories go).  Returns (mean, std, th_cell, th_seed, test-set
        hash), or None when stopped. """
        if label_type == 'distance':
            candidates = []
            for sub_dir in sorted(p for p in prediction_path.iterdir() if p.is_dir()):
                per_image = self._aji_of_directory(sub_dir, test_set_path)
                if per_image is None:
                    return None
                candidates.append((float(np.mean(list(per_image.values()))), sub_dir, per_image))
            # a strictly better mean replaces the current choice; the first directory is the fallback when every mean is
            # 0 (the reference has no winner there and fails on an unbound name)
            best_mean, best_dir, per_image = candidates[0]
            for mean, sub_dir, scores in candidates[1:]:
                if mean > best_mean:
                    best_mean, best_dir, per_image = mean, sub_dir, scores
            th_cell, th_seed = (float(part) for part in (best_dir.stem.split('_')[0], best_dir.name.split('_')[-1]))
            for file in best_dir.glob('*'):
                shutil.move(str(file), str(prediction_path / file.name))
            for _, sub_dir, _ in candidates:
                shutil.rmtree(sub_dir)
        else:
            per_image = self._aji_of_directory(prediction_path, test_set_path)
            if per_image is None:
                return None
            th_cell = th_seed = -1
        names, values = list(per_image.keys()), list(per_image.values())
        pd.DataFrame({'test image': names, 'aji+': values}).sort_values(by=['test image']).to_csv(
            prediction_path / "scores.csv", header=True, index=False)
        version = hashlib.sha1(str(names).encode("UTF-8")).hexdigest()[:10]
        return np.mean(values), np.std(values), th_cell, th_seed, version

    # -- prediction ----------------------------------------------------------------------------------------------------------
    def inference(self, net, dataset, label_type, ths, batch_size, device, path_model, save_raw, eval_progress):
        """ Predict the data set and write one mask per image and threshold pair: ``<th_cell>_<th_seed>/mask*.tif``
        (distance) or ``mask*.tif`` (boundary); with ``save_raw`` the raw prediction next to each mask. """
        if device.type == "cpu":
            raise RuntimeError("microbeseg_amd evaluation needs the MI355X HIP path (no CPU fallback)")
        try:
            workers = min(cpu_count() // 2, 16)
        except (AttributeError, NotImplementedError):
            workers = 4
        loader = torch.utils.data.DataLoader(dataset, batch_size=batch_size, shuffle=False, pin_memory=True,
                                             num_workers=getattr(self, 'num_workers', workers))
        net.eval()
        with torch.no_grad():          # the reference switches autograd off globally; here only around the loop
            for step, (images, ids, pad_lists, _sizes) in enumerate(loader):
                if step % 5 == 0:
                    QCoreApplication.processEvents()
                    if self.stop_evaluation:
                        self.text_output.emit("Stop evaluation due to user interaction.")
                        self.is_evaluating = False
                        return
                # default collate: per-sample [pad_y, pad_x] lists -> two tensors of batch size; one shape per batch
                pad_y, pad_x = int(pad_lists[0][0]), int(pad_lists[1][0])
                prediction = net(images.to(device))
                for k, image_id in enumerate(ids):
                    suffix = image_id.split('img')[-1]
                    if label_type == 'distance':
                        self._write_distance_masks(prediction, k, pad_y, pad_x, ths, path_model, suffix, save_raw)
                    else:
                        self._write_boundary_mask(prediction, k, pad_y, pad_x, path_model, suffix, save_raw)
                self.progress.emit(int(100 * (eval_progress[0] * (step + 1) * batch_size / len(dataset)
                                              + eval_progress[1])))

    @staticmethod
    def _write_distance_masks(prediction, k, pad_y, pad_x, ths, path_model, suffix, save_raw):
        border, cell = (t[k, 0, pad_y:, pad_x:].contiguous() for t in prediction)
        masks, _, _ = pp.distance_postprocessing_sweep_device(border, cell, ths, col_major_ids=True)
        masks = masks.cpu().numpy().view(np.uint16)
        raw = np.stack((cell.cpu().numpy(), border.cpu().numpy()), axis=0) if save_raw else None     # (2, H, W)
        for mask, (th_cell, th_seed) in zip(masks, ths):
            folder = path_model / "{}_{}".format(th_cell, th_seed)
            folder.mkdir(exist_ok=True)
            if save_raw:
                tiff.imwrite(str(folder / "raw{}.tif".format(suffix)), raw)
            tiff.imwrite(str(folder / "mask{}.tif".format(suffix)), mask)

    @staticmethod
    def _write_boundary_mask(logits, k, pad_y, pad_x, path_model, suffix, save_raw):
        logits = logits.contiguous()
        _, _, hp, wp = logits.shape
        probs = torch.empty((hp - pad_y, wp - pad_x, 3), dtype=torch.float32, device=logits.device)
        _lib.check(_lib.load().mseg_softmax3_hwc(logits[k].data_ptr(), hp, wp, pad_y, pad_x, probs.data_ptr(),
                                                 torch.cuda.current_stream().cuda_stream), "softmax3_hwc")
        mask, _, _ = pp.boundary_postprocessing_device(probs)
        if save_raw:
            tiff.imwrite(str(path_model / "raw{}.tif".format(suffix)), np.transpose(probs.cpu().numpy(), (2, 0, 1)))
        tiff.imwrite(str(path_model / "mask{}.tif".format(suffix)), mask.cpu().numpy().view(np.uint16))


def _is_out_of_memory(err):
    """only allocation failures end an evaluation with the reference's memory message; other errors surface"""
    text = str(err).lower()
    return "out of memory" in text or "hiperroroutofmemory" in text

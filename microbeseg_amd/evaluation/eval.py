"""Model evaluation on the test set: batched inference, threshold sweep of the post-processing, AJI+ scoring.

Mirror of ``EvalWorker`` in ``src/evaluation/eval.py`` (reference: ``start_evaluation`` :36-196, ``calc_scores``
:222-322, ``inference`` :324-425): same method names, arguments, messages, output files (``<results>/<set>_<model>/``
with ``mask*.tif``, ``scores.csv``, ``test_set.zip``; ``<results>.csv`` with one row per model) and the same selection
rule (best mean AJI+ over the 4 x 2 threshold grid of the distance method).

MI355X path (the direct caller of the hot path):
  * the network forward of a batch and the post-processing of every image stay in HBM; the 8 threshold pairs of an image
    are ONE call (``mseg_distance_postprocess_sweep``: the smoothed cell map is shared), only the uint16 masks return;
  * scoring: border correction + connected-component relabelling + area / intersection counts on the device
    (``mseg_eval_relabel``, ``mseg_eval_pair_counts``); the Hungarian pairing on the small IoU matrix is scipy on the
    host, as in the reference.
``num_gpus`` is accepted for signature compatibility; a worker evaluates on its own GPU (one process per GPU).
"""
import gc
import hashlib
import json
import os
import shutil
import zipfile
from copy import deepcopy
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


class EvalWorker(QObject):
    """ Worker class for model evaluation """
    finished = pyqtSignal()
    progress = pyqtSignal(int)
    text_output = pyqtSignal(str)
    stop_evaluation = False
    is_evaluating = False

    def start_evaluation(self, path_data, path_results, models, batch_size, device, num_gpus, save_raw_pred,
                         start_message=''):
        """ Evaluate ``models`` (paths of ``.pth`` checkpoints with their ``.json``) on ``path_data / 'test'``. """
        if not path_data.is_dir():
            self.progress.emit(0)
            self.finished.emit()
            return
        if len(list((path_data / 'test').glob('mask*'))) < 2:
            self.text_output.emit('Not enough test images found. At least 2 are needed (better more)')
            self.progress.emit(0)
            self.finished.emit()
            return
        self.is_evaluating = True
        self.text_output.emit(start_message)
        trainset_scores = {'model': [], 'th_cell': [], 'th_seed': [], 'aji+ (mean)': [], 'aji+ (std)': [],
                           'test set version': []}
        for i, model in enumerate(models):
            model_dir = path_results / "{}_{}".format(model.parent.stem, model.stem)
            if not path_results.is_dir():
                path_results.mkdir()
            if model_dir.is_dir():
                shutil.rmtree(model_dir)
            model_dir.mkdir()

            QCoreApplication.processEvents()
            if self.stop_evaluation:
                if self.is_evaluating:
                    self.text_output.emit("Stop evaluation due to user interaction.")
                break

            with open(model.parent / "{}.json".format(model.stem)) as f:
                model_settings = json.load(f)
            arch = model_settings['architecture']
            net = build_unet(unet_type=arch[0], act_fun=arch[2], pool_method=arch[1], normalization=arch[3],
                             device=device, num_gpus=1, ch_in=1,
                             ch_out=1 if model_settings['label_type'] == 'distance' else 3, filters=arch[4])
            net = get_weights(net=net, weights=str(model), num_gpus=1, device=device)
            test_dataset = InferenceDataset(data_dir=path_data / 'test',
                                            transform=pre_processing_transforms(apply_clahe=False, scale_factor=1))
            if model_settings['label_type'] == 'distance':
                th_cell, th_seed = [0.05, 0.075, 0.10, 0.125], [0.35, 0.45]
                ths = list(product(th_cell, th_seed))
            else:
                ths = [-1]
            try:
                self.inference(net=net, dataset=test_dataset, label_type=model_settings['label_type'], ths=ths,
                               batch_size=batch_size, device=device, path_model=model_dir, save_raw=save_raw_pred,
                               eval_progress=(0.5 / len(models), i / len(models)))
            except RuntimeError:
                text = "Please, try again with smaller batch size or reduce the crop size (use the export " \
                       "and import functionalities for this)"
                self.text_output.emit(text)
                self.text_output.emit('Stop evaluation due to memory problems')
                self.text_output.emit(text)
                self.finished.emit()
                return
            del net
            gc.collect()

            results = self.calc_scores(prediction_path=model_dir, test_set_path=path_data / 'test',
                                       label_type=model_settings['label_type'])
            if results:
                trainset_scores['model'].append("{}: {}".format(model.parent.stem, model.stem))
                trainset_scores['th_cell'].append(results[2])
                trainset_scores['th_seed'].append(results[3])
                trainset_scores['aji+ (mean)'].append(results[0])
                trainset_scores['aji+ (std)'].append(results[1])
                trainset_scores['test set version'].append(results[4])
                with zipfile.ZipFile(model_dir / 'test_set.zip', 'w') as z:
                    z.write(path_data, arcname=path_data.stem, compress_type=zipfile.ZIP_DEFLATED)
                    z.write(path_data / 'test', arcname=os.path.join(path_data.stem, 'test'),
                            compress_type=zipfile.ZIP_DEFLATED)
                    for file in (path_data / 'test').glob('*'):
                        z.write(file, arcname=os.path.join(path_data.stem, 'test', file.name),
                                compress_type=zipfile.ZIP_DEFLATED)
            self.progress.emit(int(100 * (i + 1) / len(models)))

        if not self.stop_evaluation and trainset_scores['model']:
            trainset_scores_df = pd.DataFrame(trainset_scores)
            csv_path = path_results.parent / '{}.csv'.format(path_results.stem)
            if csv_path.is_file():
                old = pd.read_csv(csv_path)
                old = old[old['test set version'] == trainset_scores_df.iloc[0]['test set version']]
                trainset_scores_df = pd.concat([trainset_scores_df, old])     # DataFrame.append of the reference
                trainset_scores_df = trainset_scores_df.drop_duplicates('model')
            trainset_scores_df = trainset_scores_df.sort_values(by=['model'])
            trainset_scores_df.to_csv(csv_path, header=True, index=False)
            self.progress.emit(100)
        self.is_evaluating = False
        self.finished.emit()
        return

    @pyqtSlot()
    def stop_evaluation_process(self):
        """ Set internal evaluation stop state to True """
        self.stop_evaluation = True

    def _score_dir(self, directory, test_set_path):
        scores, file_names = [], []
        for pred_id in directory.glob('mask*.tif'):
            QCoreApplication.processEvents()
            if self.stop_evaluation:
                if self.is_evaluating:
                    self.text_output.emit("Stop metric calculation.")
                return None, None
            prediction = tiff.imread(str(pred_id))
            ground_truth = tiff.imread(str(test_set_path / pred_id.name))
            # border_correction + measure.label + get_fast_aji_plus (eval.py:248-256), fused on the device
            scores.append(aji_plus_masks(prediction, ground_truth))
            file_names.append(pred_id.stem)
        return scores, file_names

    def calc_scores(self, prediction_path, test_set_path, label_type):
        """ Aggregated Jaccard index AJI+ per test image; for the distance method the sub-directory (threshold pair) with
        the best mean is kept.  Returns (mean, std, th_cell, th_seed, test-set hash) or None when stopped. """
        if label_type == 'distance':
            score, score_std, th_cell, th_seed, best_sub_dir = 0, 0, 0, 0, ''
            scores, file_names = [], []
            for sub_dir in sorted(p for p in prediction_path.iterdir() if p.is_dir()):
                sub_dir_scores, names = self._score_dir(sub_dir, test_set_path)
                if sub_dir_scores is None:
                    return None
                file_names = names
                sub_dir_score = np.mean(sub_dir_scores)
                sub_dir_score_std = np.std(sub_dir_scores)
                if sub_dir_score > score or best_sub_dir == '':
                    # strictly better mean wins, like the reference; the first directory is the fallback when every
                    # score is 0 (the reference would fail with an unbound `scores` there)
                    score = sub_dir_score
                    score_std = sub_dir_score_std
                    th_cell = float(sub_dir.stem.split('_')[0])
                    th_seed = float(sub_dir.name.split('_')[-1])
                    scores = deepcopy(sub_dir_scores)
                    best_sub_dir = sub_dir.name
            for sub_dir in [p for p in prediction_path.iterdir() if p.is_dir()]:
                if sub_dir.name == best_sub_dir:
                    for f in sub_dir.glob('*'):
                        shutil.move(str(f), str(sub_dir.parents[0] / f.name))
                shutil.rmtree(sub_dir)
        else:
            scores, file_names = self._score_dir(prediction_path, test_set_path)
            if scores is None:
                return None
            score, score_std, th_cell, th_seed = np.mean(scores), np.std(scores), -1, -1
        results_df = pd.DataFrame({'test image': file_names, 'aji+': scores})
        results_df = results_df.sort_values(by=['test image'])
        results_df.to_csv(prediction_path / "scores.csv", header=True, index=False)
        return score, score_std, th_cell, th_seed, hashlib.sha1(str(file_names).encode("UTF-8")).hexdigest()[:10]

    def inference(self, net, dataset, label_type, ths, batch_size, device, path_model, save_raw, eval_progress):
        """ Predict the data set and write one mask per image and threshold pair (``<th_cell>_<th_seed>/mask*.tif``
        for the distance method, ``mask*.tif`` for the boundary method). """
        if device.type == "cpu":
            raise RuntimeError("microbeseg_amd evaluation needs the MI355X HIP path (no CPU fallback)")
        try:
            num_workers = cpu_count() // 2
        except AttributeError:
            num_workers = 4
        num_workers = int(np.minimum(num_workers, 16))
        dataloader = torch.utils.data.DataLoader(dataset, batch_size=batch_size, shuffle=False, pin_memory=True,
                                                 num_workers=getattr(self, 'num_workers', num_workers))
        lib = _lib.load()
        net.eval()
        # the reference switches autograd off globally here (eval.py:356); this build restores the caller's mode
        grad_mode = torch.is_grad_enabled()
        torch.set_grad_enabled(False)
        try:
            self._inference_loop(lib, net, dataloader, dataset, label_type, ths, batch_size, device, path_model, save_raw,
                                 eval_progress)
        finally:
            torch.set_grad_enabled(grad_mode)

    def _inference_loop(self, lib, net, dataloader, dataset, label_type, ths, batch_size, device, path_model, save_raw,
                        eval_progress):
        for i, sample in enumerate(dataloader):
            if i % 5 == 0:
                QCoreApplication.processEvents()
                if self.stop_evaluation:
                    self.text_output.emit("Stop evaluation due to user interaction.")
                    self.is_evaluating = False
                    return
            img_batch, ids_batch, pad_batch, img_size = sample
            img_batch = img_batch.to(device)
            # default collate turns the per-sample [pad_y, pad_x] lists into two tensors of batch size; all images of a
            # batch share shape and pads (eval.py:374-375)
            pads = [int(pad_batch[0][0]), int(pad_batch[1][0])]
            if label_type == 'distance':
                border_batch, cell_batch = net(img_batch)
            else:
                logits_batch = net(img_batch).contiguous()
            for h in range(len(img_batch)):
                file_id = ids_batch[h].split('img')[-1]
                if label_type == 'distance':
                    cell = cell_batch[h, 0, pads[0]:, pads[1]:].contiguous()
                    border = border_batch[h, 0, pads[0]:, pads[1]:].contiguous()
                    labels, _, _ = pp.distance_postprocessing_sweep_device(border, cell, ths, col_major_ids=True)
                    labels = labels.cpu().numpy().view(np.uint16)
                    for k, th in enumerate(ths):
                        path_results = path_model / "{}_{}".format(th[0], th[1])
                        if not path_results.is_dir():
                            path_results.mkdir()
                        if save_raw:
                            raw_pred = np.stack((cell.cpu().numpy(), border.cpu().numpy()), axis=0)   # (2, H, W)
                            tiff.imwrite(str(path_results / "raw{}.tif".format(file_id)), raw_pred)
                        tiff.imwrite(str(path_results / "mask{}.tif".format(file_id)), labels[k])
                else:
                    _, _, hp, wp = logits_batch.shape
                    probs = torch.empty((hp - pads[0], wp - pads[1], 3), dtype=torch.float32, device=device)
                    _lib.check(lib.mseg_softmax3_hwc(logits_batch[h].data_ptr(), hp, wp, pads[0], pads[1],
                                                     probs.data_ptr(), torch.cuda.current_stream().cuda_stream),
                               "softmax3_hwc")
                    labels, _, _ = pp.boundary_postprocessing_device(probs)
                    if save_raw:
                        tiff.imwrite(str(path_model / "raw{}.tif".format(file_id)),
                                     np.transpose(probs.cpu().numpy(), (2, 0, 1)))
                    tiff.imwrite(str(path_model / "mask{}.tif".format(file_id)), labels.cpu().numpy().view(np.uint16))
            self.progress.emit(int(100 * (eval_progress[0] * (i + 1) * batch_size / len(dataset) + eval_progress[1])))
        return

"""Inference driver of the hot path: per frame normalise -> U-Net forward (eval) -> un-pad -> post-processing.

Mirror of ``InferWorker`` in ``src/inference/infer.py`` (reference: constructor :30-111, ``start_inference`` :113-326,
``inference`` :328-376) and of the frame loop of ``infer_script_local.py`` (:118-161).  The OMERO plumbing of
``start_inference`` (login, plane download, ROI upload) is outside the hot path; what is
kept is the per-frame contract — ``inference(img, min_val, max_val, pads)`` with an already padded frame, returning the
``np.uint16`` instance mask of the un-padded frame — plus ``infer_stack`` for local [T, H, W] stacks.
On the MI355X path the frame stays in HBM between the network and the watershed; only the uint16 mask returns.
"""
import json
from pathlib import Path

import numpy as np
import torch

from .. import _lib, engine
from ..utils.qt_shim import QObject, pyqtSignal, pyqtSlot
from ..utils.unets import build_unet, get_weights
from ..utils.utils import zero_pad_model_input
from . import postprocessing as pp


def is_out_of_memory(err):
    """The reference turns EVERY RuntimeError of ``self.net(img_batch)`` into an all-zero mask (infer.py:351-356: "maybe
    not enough ram/vram?").  Here only a real allocation failure of the network forward gets that treatment; a kernel
    launch error of libmseg_hip, an asynchronous HIP fault or a shape bug propagates instead of silently writing empty
    segmentations."""
    msg = str(err).lower()
    return isinstance(err, (RuntimeError, MemoryError)) and (
        "out of memory" in msg or "hiperroroutofmemory" in msg or "hip error: out of memory" in msg)


def load_model(model, device):
    """``model``: path of the checkpoint without/with suffix; reads ``<model>.json`` (architecture, label_type) and
    ``<model>.pth`` (state dict), like infer.py:83-84,119-131.  Returns (net in eval mode, model_settings)."""
    model = Path(model)
    base = model.parent / model.stem
    if not base.with_suffix('.pth').is_file():
        raise Exception(f'{base.with_suffix(".pth")} not found!')
    if not base.with_suffix('.json').is_file():
        raise Exception(f'{base.with_suffix(".json")} not found!')
    with open(base.with_suffix('.json')) as f:
        settings = json.load(f)
    arch = settings['architecture']
    net = build_unet(unet_type=arch[0], act_fun=arch[2], pool_method=arch[1], normalization=arch[3], device=device,
                     num_gpus=1, ch_in=1, ch_out=1 if settings['label_type'] == 'distance' else 3, filters=arch[4])
    net = get_weights(net=net, weights=str(base.with_suffix('.pth')), num_gpus=1, device=device)
    net.eval()
    return net, settings


class InferWorker(QObject):
    """ Worker class for inference """
    finished = pyqtSignal()
    progress = pyqtSignal(int)
    text_output = pyqtSignal(str)
    stop_inference = False
    # [extension] "bf16": the network forward runs with bf16 matrix-core inputs (fp32 accumulate / storage),
    # ~3x faster; predictions differ from the fp32 reference arithmetic in the third digit, so masks are no longer
    # guaranteed bit-identical to the reference's — opt-in only, the default is the reference's fp32
    precision = "fp32"
    # [extension] edge length of the tiles of sliding-window inference (``sliding_window=True``; inference/tiling.py)
    tile_size = 2048
    # [measurement hook, None in production] callable(prediction) -> prediction, applied to the network's output before the
    # post-processing.  bench.py uses it to hand the watershed realistic distance maps: an UNTRAINED network (there are no
    # checkpoints offline) predicts one confluent blob, whose flood is a single sequential component.
    prediction_hook = None
    BOUNDARY_BATCH = 8      # infer_stack, boundary method: frames whose floods share one launch (1..8)

    def __init__(self, img_id_list=None, inference_path=None, omero_username=None, omero_password=None, omero_host=None,
                 omero_port=None, group_id=None,
                 model=None, device='cuda:0', ths=(0.10, 0.45), channel=0, upload=True, overwrite=True,
                 sliding_window=False, print_output=False):
        super().__init__()
        self.img_id_list = img_id_list
        self.inference_path = inference_path
        self.omero_username, self.omero_password = omero_username, omero_password   # names: reference infer.py:30,66-69
        self.omero_host, self.omero_port, self.group_id = omero_host, omero_port, group_id
        self.model = model
        self.device = torch.device(device)
        self.ths = list(ths)           # [th_cell, th_seed] (infer.py:362-365)
        self.channel = channel
        self.upload = upload
        self.overwrite = overwrite
        # the reference stores this flag and never reads it (infer.py:60,76); here it switches on tiled inference, whose
        # prediction equals whole-frame inference (inference/tiling.py) and which lifts the 8192-px limit of the padding
        self.sliding_window = sliding_window
        self.print_output = print_output
        self.net, self.model_settings = (None, None)
        if model is not None:
            self.net, self.model_settings = load_model(model, self.device)

    def start_inference(self):
        """The reference pulls planes from an OMERO server here (infer.py:113-326): not part of this build."""
        raise RuntimeError("InferWorker.start_inference needs the OMERO stack (omero-py), which is outside the "
                           "MI355X hot path; use infer_stack()/inference() or infer_script_local.py")

    def pad_frame(self, img_frame, pad_val):
        """top / left padding of a frame up to the next tested shape, like the reference (utils.py:124-163).  Frames beyond
        8192 px raise 'Image too big to pad. Use sliding windows' there; with ``sliding_window`` they are padded to the
        16-px grid of the network instead (smaller frames keep the reference's padding, so that tiled and whole-frame
        inference see the same input)."""
        if self.sliding_window and max(img_frame.shape[:2]) > 8192:
            from .tiling import pad_to_grid
            return pad_to_grid(img_frame, pad_val)
        return zero_pad_model_input(img_frame, pad_val=pad_val)

    def inference(self, img, min_val, max_val, pads):
        """ Predict one (already padded) frame.

        :param img: padded frame (2-D numpy array, any integer/float dtype).
        :param min_val: minimum of the un-padded frame (numpy scalar of the image dtype).
        :param max_val: maximum of the un-padded frame.
        :param pads: [rows padded at the top, columns padded at the left] (removed after the forward pass).
        :return: instance mask, np.uint16, shape of the un-padded frame.
        """
        self.net.eval()
        # 2 * (f32(img) - min) / (max - min) - 1, same operation order and scalar types as infer.py:346-348
        img_batch = 2 * (img.astype(np.float32) - min_val) / (max_val - min_val) - 1
        img_batch = torch.from_numpy(np.ascontiguousarray(img_batch[None, None, :, :])).to(torch.float)
        with torch.no_grad():   # the reference disables autograd globally (infer.py:343); here only for the call
            pred = self._forward(img_batch)
            if pred is None:        # zero mask instead of a crash (infer.py:354-356) — out-of-memory only
                return np.zeros_like(img, dtype=np.uint16)[pads[0]:, pads[1]:]
            return self._postprocess(pred, pads).cpu().numpy().view(np.uint16)

    def _forward(self, img_batch):
        """network forward of one padded frame; None (after the reference's message) if it does not fit in memory"""
        try:
            with engine.precision_scope(self.precision):
                if isinstance(img_batch, engine.RawFrame):      # raw frame on the device: normalised by the first kernel
                    return self.net(img_batch)
                if self.sliding_window:
                    from .tiling import tiled_forward
                    return tiled_forward(self.net, img_batch.to(self.device), tile=self.tile_size)
                return self.net(img_batch.to(self.device))
        except (RuntimeError, MemoryError) as err:
            if not is_out_of_memory(err):
                raise
            self.text_output.emit('RuntimeError during inference (maybe not enough ram/vram?)')
            return None

    def _postprocess(self, pred, pads):
        """prediction (device) -> uint16 labels of the un-padded frame (device tensor, int16 storage)"""
        lib = _lib.load()
        if self.model_settings['label_type'] == 'distance':
            border, cell = pred
            cell = cell[0, 0, pads[0]:, pads[1]:].contiguous()
            border = border[0, 0, pads[0]:, pads[1]:].contiguous()
            # every reference caller hands (H, W, 1) arrays to distance_postprocessing -> column-major instance ids
            labels, _, _ = pp.distance_postprocessing_device(border, cell, th_seed=self.ths[1], th_cell=self.ths[0],
                                                             col_major_ids=True)
        else:
            labels, _, _ = pp.boundary_postprocessing_device(self._softmax_hwc(pred, pads))
        return labels

    @staticmethod
    def _softmax_hwc(pred, pads):
        """(1, 3, Hp, Wp) logits -> (H, W, 3) softmax probabilities of the un-padded frame (device, current stream)"""
        lib = _lib.load()
        logits = pred.contiguous()
        _, _, hp, wp = logits.shape
        probs = torch.empty((hp - pads[0], wp - pads[1], 3), dtype=torch.float32, device=logits.device)
        _lib.check(lib.mseg_softmax3_hwc(logits.data_ptr(), hp, wp, int(pads[0]), int(pads[1]), probs.data_ptr(),
                                         torch.cuda.current_stream().cuda_stream), "softmax3_hwc")
        return probs

    def infer_stack(self, img):
        """[T, H, W] stack -> [T, H, W] uint16 masks; per frame min/max + top/left padding exactly like
        infer_script_local.py:118-161 / infer.py:250-259.

        MI355X path for distance models: the frames are pipelined over two HIP streams — while the watershed of frame i
        (a few long-running, latency-bound lanes) runs on the side stream, the matrix kernels of frame i+1 run on the
        main stream, and the uint16 mask travels back through a pinned buffer.  Results are identical to calling
        ``inference`` frame by frame."""
        results = np.zeros(shape=(img.shape[0], img.shape[1], img.shape[2]), dtype=np.uint16)
        pipelined = (self.model_settings is not None and self.model_settings['label_type'] in ('distance', 'boundary')
                     and self.device.type == 'cuda')
        boundary = pipelined and self.model_settings['label_type'] == 'boundary'
        if not pipelined:
            for frame in range(len(img)):
                if self.stop_inference:
                    break
                img_frame = np.copy(img[frame])
                frame_min, frame_max = np.min(img_frame), np.max(img_frame)
                img_frame, pads = self.pad_frame(img_frame, frame_min)
                results[frame] = self.inference(img_frame, frame_min, frame_max, pads)
                self.progress.emit(int(100 * (frame + 1) / len(img)))
            return results

        self.net.eval()
        # Side streams for the post-processing.  Distance method: one (the watershed of frame i under the network of frame
        # i + 1).  Boundary method: its flood is ONE wavefront busy for ~45 ms per 2048^2 frame (the heap's
        # marker phase is sequential by definition) — a latency, not a load.  The frames of a stack are collected in groups of
        # BOUNDARY_BATCH and a group's floods go into ONE launch, one workgroup per frame (mseg_boundary_flood_batch: eight
        # floods take the time of one); two side streams alternate between groups, each group on its own workspace slots,
        # while the network goes on with the next frames.  HIGH-priority streams: HIP multiplexes streams onto a few hardware
        # queues per priority level and kernels of streams that share a queue run one after the other — at the default
        # priority a 45-ms flood sat in front of the network's kernels whenever its stream shared the main stream's queue.
        batch = max(1, min(int(self.BOUNDARY_BATCH), 8)) if boundary else 1
        if boundary:      # two groups' workspaces (0.32 GiB per 2048^2 frame, 5.2 GiB per 8192^2 frame) stay below ~16 GiB
            per_frame = max(1, _lib.load().mseg_postproc_workspace_bytes(int(img.shape[1]), int(img.shape[2])))
            batch = max(1, min(batch, (8 << 30) // per_frame))
        nside = 2 if boundary else 1
        sides = [torch.cuda.Stream(device=self.device, priority=-1) if boundary else torch.cuda.Stream(device=self.device)
                 for _ in range(nside)]
        in_flight = 2 * batch if boundary else 2
        pending = []      # (frame index, pinned host mask, event on the side stream)
        group, groups_done = [], [0]

        def finish(entry):
            f, host, ev = entry
            if ev is not None:
                ev.synchronize()
                results[f] = host.numpy().view(np.uint16)
            self.progress.emit(int(100 * (f + 1) / len(img)))

        def flush_group():
            """boundary method: softmax + post-processing of the collected frames on one side stream, their floods in one launch"""
            if not group:
                return
            side = sides[groups_done[0] % nside]
            slot0 = (groups_done[0] % nside) * 8
            groups_done[0] += 1
            ready = torch.cuda.Event()
            ready.record()
            with torch.cuda.stream(side):
                side.wait_event(ready)
                outs = pp.boundary_postprocessing_batch_device([self._softmax_hwc(lg, pads) for _, lg, pads in group],
                                                               first_slot=slot0)
                for (frame, lg, _), (labels, _, _) in zip(group, outs):
                    lg.record_stream(side)
                    host = torch.empty(labels.shape, dtype=torch.int16, pin_memory=True)
                    host.copy_(labels, non_blocking=True)
                    done = torch.cuda.Event()
                    done.record(side)
                    pending.append((frame, host, done))
            group.clear()

        def launch_postproc(frame, pred, pads):
            if self.prediction_hook is not None:
                pred = self.prediction_hook(pred)
            if boundary:
                group.append((frame, pred.contiguous(), pads))
                if len(group) >= batch:
                    flush_group()
                return
            side = sides[0]
            border, cell = pred
            cell = cell[0, 0, pads[0]:, pads[1]:].contiguous()
            border = border[0, 0, pads[0]:, pads[1]:].contiguous()
            ready = torch.cuda.Event()
            ready.record()
            with torch.cuda.stream(side):
                side.wait_event(ready)
                labels, _, _ = pp.distance_postprocessing_device(border, cell, th_seed=self.ths[1],
                                                                 th_cell=self.ths[0], col_major_ids=True)
                border.record_stream(side)
                cell.record_stream(side)
                host = torch.empty(labels.shape, dtype=torch.int16, pin_memory=True)
                host.copy_(labels, non_blocking=True)
                done = torch.cuda.Event()
                done.record(side)
            pending.append((frame, host, done))

        # K14 on the device (uint8 / uint16 stacks, whole-frame inference): the raw frame goes up from a pinned staging
        # buffer on a copy stream, its extrema are reduced on the device and the first convolution normalises / pads while it
        # loads (engine.RawFrame) — the host loop only copies into pinned memory and enqueues.  Same input values bit for
        # bit as the host formula of inference() (tests/test_gpu_fullsize.py), so the masks are identical.
        device_norm = (img.dtype in (np.uint8, np.uint16) and not self.sliding_window and img.shape[1] <= 8192
                       and img.shape[2] <= 8192)
        with torch.cuda.device(self.device), torch.no_grad():
            if device_norm:
                from ..utils.utils import pad_amounts
                pads = pad_amounts(img.shape[1:])
                tdt = torch.uint8 if img.dtype == np.uint8 else torch.int16      # (uint16 bits in int16 storage)
                stage = [torch.empty(img.shape[1:], dtype=tdt, pin_memory=True) for _ in range(2)]
                raw_dev = [torch.empty(img.shape[1:], dtype=tdt, device=self.device) for _ in range(2)]
                uploaded = [None, None]          # copy-stream events: staging buffer k has left the host
                consumed = [None, None]          # main-stream events: the network has read device buffer k
                copy_stream = torch.cuda.Stream(device=self.device)
                main = torch.cuda.current_stream()
            for frame in range(len(img)):
                if self.stop_inference:
                    break
                if device_norm:
                    k = frame & 1
                    if uploaded[k] is not None:
                        uploaded[k].synchronize()            # the staging buffer is free again
                    np.copyto(stage[k].numpy().view(img.dtype), img[frame])
                    with torch.cuda.stream(copy_stream):
                        if consumed[k] is not None:
                            copy_stream.wait_event(consumed[k])
                        raw_dev[k].copy_(stage[k], non_blocking=True)
                        uploaded[k] = torch.cuda.Event()
                        uploaded[k].record(copy_stream)
                    main.wait_event(uploaded[k])
                    pred = self._forward(engine.RawFrame(raw_dev[k], pads[0], pads[1]))
                    consumed[k] = torch.cuda.Event()
                    consumed[k].record(main)
                else:
                    img_frame = np.copy(img[frame])
                    frame_min, frame_max = np.min(img_frame), np.max(img_frame)
                    img_frame, pads = self.pad_frame(img_frame, frame_min)
                    img_batch = 2 * (img_frame.astype(np.float32) - frame_min) / (frame_max - frame_min) - 1
                    img_batch = torch.from_numpy(np.ascontiguousarray(img_batch[None, None, :, :])).to(torch.float)
                    pred = self._forward(img_batch)
                if pred is None:                 # out of memory: zero mask, like inference() (infer.py:354-356)
                    pending.append((frame, None, None))
                else:
                    launch_postproc(frame, pred, pads)
                while len(pending) > in_flight:  # distance: two frames in flight; boundary: two groups
                    finish(pending.pop(0))
            flush_group()
            while pending:
                finish(pending.pop(0))
        return results

    # -- what follows the prediction on the reference's routes (infer.py:265-291 upload, :320-322 local save) -------------
    STROKE_COLOR = int.from_bytes([255, 255, 0, 255], byteorder='big', signed=True)     # yellow, opaque (infer.py:271-272)

    def polygon_rois(self, prediction, frame=0):
        """ One polygon ROI per cell of a predicted frame, as plain records with the fields the reference sets on its
        ``omero.model.PolygonI`` objects (theZ, theT, theC, fillColor, strokeColor, points = "x,y x,y ... ").  The
        contours of ALL instances are traced on the device in two launches (utils/hull_polygon.py) instead of one
        ``cv2.findContours`` call per instance. """
        from ..utils.hull_polygon import label_polygons, points_string
        rois = []
        if np.max(prediction) > 0:
            for polygons in label_polygons(prediction).values():
                for polygon in polygons:
                    rois.append({'theZ': 0, 'theT': int(frame), 'theC': int(self.channel), 'fillColor': 0,
                                 'strokeColor': self.STROKE_COLOR, 'points': points_string(polygon)})
        return rois

    def save_results(self, results_array, image_name, result_path=None, with_rois=False):
        """ ``<result_path>/<image stem>_channel<c>.tif`` (uint16 [T, H, W], the reference's local-save route) and, with
        ``with_rois``, ``<...>_rois.json``: the polygon ROIs per frame that the upload route would send to OMERO. """
        from ..utils import tiffio as tiff
        result_path = Path(self.inference_path if result_path is None else result_path)
        result_path.mkdir(parents=True, exist_ok=True)
        stem = Path(image_name).stem
        target = result_path / f"{stem}_channel{self.channel}.tif"
        tiff.imwrite(str(target), results_array)
        if with_rois:
            frames = results_array if results_array.ndim == 3 else results_array[None]
            rois = [roi for t, frame in enumerate(frames) for roi in self.polygon_rois(frame, t)]
            with open(result_path / f"{stem}_channel{self.channel}_rois.json", 'w', encoding='utf-8') as f:
                json.dump({'image': str(image_name), 'channel': int(self.channel), 'rois': rois}, f)
        return target

    @pyqtSlot()
    def inference_finished(self):
        self.finished.emit()

    @pyqtSlot()
    def stop_inference_process(self):
        """ Set internal stop state to True """
        self.stop_inference = True

#!/usr/bin/env python3
"""Segment local TIFF images / 2D+t stacks with a trained model on the MI355X hot path.

Same flags and output as the reference ``infer_script_local.py`` (:17-25, :164-165): ``--img_dir/-i``, ``--model/-m``,
``--thresholds/-t`` (th_cell th_seed, default 0.10 0.45), ``--result_path/-r``, ``--channel/-c``, ``--device/-d``,
``--overwrite/-o``; writes ``mask_<stem>_channel<c>.tif`` (uint16, [T, H, W] squeezed).
"""
import argparse
from pathlib import Path

import numpy as np
import torch


def select_frames(img, channel, name):
    """-> [T, H, W] (same rules as infer_script_local.py:85-101); None = unsupported shape."""
    if img.ndim == 2:
        return img[None, ...]
    if img.ndim == 3:
        if img.shape[-1] == 3:
            return img[..., channel][None, ...]
        if img.shape[0] == 3:
            return img[channel, ...][None, ...]
        return img
    if img.ndim == 4:
        return img[:, channel, ...]
    if img.ndim == 5:
        print(f'Skip {name} (not supported image shape)')
        return None
    raise Exception('Adapt script for your data format!')


def main():
    parser = argparse.ArgumentParser(description='microbeSEG inference on local files (MI355X-native hot path)')
    parser.add_argument('--img_dir', '-i', required=True, type=str, help='Directory with .tif images / stacks')
    parser.add_argument('--model', '-m', required=True, type=str, help='Model to use (path without suffix)')
    parser.add_argument('--thresholds', '-t', default=[0.10, 0.45], nargs='+', type=float,
                        help='Thresholds for distance method: th_cell th_seed')
    parser.add_argument('--result_path', '-r', default=None, type=str, help='Path for saving results')
    parser.add_argument('--channel', '-c', default=0, type=int, help='Channel to segment')
    parser.add_argument('--device', '-d', default='cuda:0', type=str, help='"cuda:N"')
    parser.add_argument('--overwrite', '-o', default=False, action='store_true', help='Overwrite existing results')
    parser.add_argument('--precision', default='fp32', choices=['fp32', 'bf16'],
                        help='[extension] bf16 = bf16 matrix-core inputs for the network (faster; masks are no longer '
                             'guaranteed identical to the fp32 reference arithmetic)')
    parser.add_argument('--sliding_window', default=False, action='store_true',
                        help='[extension] tiled inference (2048 px tiles + 128 px halo; same prediction as whole-frame '
                             'inference for the BatchNorm models training produces); lifts the 8192 px frame limit')
    parser.add_argument('--rois', default=False, action='store_true',
                        help='[extension] also write <mask file stem>_rois.json: one polygon ROI per cell and frame, the '
                             'records the OMERO route of infer_script.py uploads (traced on the device)')
    args = parser.parse_args()

    imgs_path = Path(args.img_dir)
    result_path = (Path(__file__).parent / 'results') if args.result_path is None else Path(args.result_path)
    result_path.mkdir(exist_ok=True)
    if len(args.thresholds) != 2:
        raise Exception(f"{len(args.thresholds)} threshold given, needed are 2")
    if 'cuda' in args.device and not torch.cuda.is_available():
        raise ValueError('No MI355X visible: this build has no CPU inference path')

    from microbeseg_amd.inference.infer import InferWorker
    from microbeseg_amd.utils import tiffio as tiff

    file_ids = sorted(imgs_path.glob('*.tif*'))
    if len(file_ids) == 0:
        print('No files found')
        return
    worker = InferWorker(model=args.model, device=args.device, ths=args.thresholds, channel=args.channel,
                         sliding_window=args.sliding_window)
    worker.precision = args.precision
    worker.text_output.connect(print)
    torch.set_grad_enabled(False)
    print('--- Start inference ---')
    for img_id in file_ids:
        out_file = result_path / f"mask_{img_id.stem}_channel{args.channel}.tif"
        frames = select_frames(tiff.imread(str(img_id)), args.channel, img_id.name)
        if frames is None:
            continue
        if out_file.is_file() and not args.overwrite:
            print(f'Skip {img_id.stem} (already processed and overwriting not enabled)')
            continue
        print(f'Process {img_id.stem} (channel: {args.channel})')
        results = worker.infer_stack(frames)
        tiff.imwrite(str(out_file), np.squeeze(results))
        if args.rois:
            import json
            rois = [roi for t in range(len(results)) for roi in worker.polygon_rois(results[t], t)]
            with open(out_file.with_name(out_file.stem + '_rois.json'), 'w', encoding='utf-8') as f:
                json.dump({'image': img_id.name, 'channel': args.channel, 'rois': rois}, f)
    print('--- Finished ---')


if __name__ == "__main__":
    main()

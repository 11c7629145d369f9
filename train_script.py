#!/usr/bin/env python3
"""Train microbeSEG models from the command line on the MI355X hot path.

Same flags as the reference ``train_script.py`` (:16-28): ``--omero_id/-id``, ``--batch_size/-b`` (4),
``--iterations/-i`` (1), ``--method/-m`` (distance), ``--optimizer/-o`` (Ranger), ``--model_path/-r``,
``--device/-d`` (cuda:0), OMERO credentials.  Extensions: ``--train_path`` (a local, already exported + labelled
training set ``<path>/{train,val}/{img,mask,cell_dist,neighbor_dist|boundary}_*.tif``),
``--filters F0 F1`` and ``--max_epochs``.  Exporting a set from OMERO and creating the training labels
(reference train_script.py:41-114) are outside the hot path and need the reference's OMERO stack.
Multi-GPU: ``python -m torch.distributed.run --nproc-per-node N train_script.py ...`` (one process per GPU, RCCL).
"""
import argparse
import os
from pathlib import Path

import torch


def main():
    parser = argparse.ArgumentParser(description='Train microbeSEG models (MI355X-native hot path)')
    parser.add_argument('--omero_id', '-id', default=None, type=int, help='Omero training dataset id')
    parser.add_argument('--batch_size', '-b', default=4, type=int, help='Batch size (per GPU)')
    parser.add_argument('--iterations', '-i', default=1, type=int, help='Number of models to train')
    parser.add_argument('--method', '-m', default='distance', type=str, help='"boundary" or "distance" method')
    parser.add_argument('--optimizer', '-o', default='Ranger', type=str, help='"Ranger" or "Adam" optimizer')
    parser.add_argument('--model_path', '-r', default=None, type=str, help='Path to save models in')
    parser.add_argument('--device', '-d', default='cuda:0', type=str, help='"cuda:N" (ROCm keeps the cuda name)')
    parser.add_argument('--username', default=None, type=str)
    parser.add_argument('--password', default=None, type=str)
    parser.add_argument('--host', default=None, type=str)
    parser.add_argument('--port', default=None, type=int)
    parser.add_argument('--train_path', default=None, type=str, help='[extension] local labelled training set')
    parser.add_argument('--filters', default=None, type=int, nargs=2, help='[extension] first / max feature maps')
    parser.add_argument('--max_epochs', default=None, type=int, help='[extension] epoch budget override')
    parser.add_argument('--precision', default='fp32', choices=['fp32', 'bf16'],
                        help='[extension] bf16 = bf16 matrix-core inputs, fp32 accumulate / storage / statistics')
    args = parser.parse_args()

    if args.method not in ('boundary', 'distance'):
        raise Exception('Unknown method. Use "boundary" or "distance"')
    if args.optimizer.lower() not in ('ranger', 'adam'):
        raise Exception('Unknown optimizer. Use "ranger" or "adam"')

    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local_rank)
        args.device = f'cuda:{local_rank}'
        dist.init_process_group('nccl', device_id=torch.device(args.device))
    if 'cuda' in args.device and not torch.cuda.is_available():
        raise ValueError('No MI355X visible: this build has no CPU training path (see oracle/ for the CPU reference)')
    device = torch.device(args.device)

    if args.train_path is None:
        raise SystemExit('OMERO export (--omero_id) needs the reference OMERO stack, which is outside the MI355X hot '
                         'path; export + label the training set with the reference tools and pass --train_path')
    path_data = Path(args.train_path)
    model_path = (Path(__file__).parent / 'models') if args.model_path is None else Path(args.model_path)
    model_path = model_path / path_data.stem
    model_path.mkdir(parents=True, exist_ok=True)

    from microbeseg_amd.training.train import TrainWorker
    worker = TrainWorker()
    worker.precision = args.precision
    worker.start_training(path_data, model_path, args.method, args.iterations, args.optimizer.lower(), args.batch_size,
                          device, world, True, filters=args.filters, max_epochs=args.max_epochs)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

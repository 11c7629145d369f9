#!/usr/bin/env python3
"""OMERO-driven inference entry point (reference ``infer_script.py`` :15-30): same flags, but the OMERO transport
(plane download, ROI / mask upload, src/inference/infer.py:113-326) is outside the MI355X hot path.  With omero-py
installed, wire ``microbeseg_amd.inference.infer.InferWorker.inference`` into the reference's worker; for files on disk
use ``infer_script_local.py``."""
import argparse


def main():
    parser = argparse.ArgumentParser(description='microbeSEG inference on OMERO data (needs the reference OMERO stack)')
    parser.add_argument('--omero_ids', '-ids', required=True, nargs='+', type=int)
    parser.add_argument('--id_type', '-i', default='dataset', type=str)
    parser.add_argument('--model', '-m', required=True, type=str)
    parser.add_argument('--thresholds', '-t', default=[0.10, 0.45], nargs='+', type=float)
    parser.add_argument('--result_path', '-r', default=None, type=str)
    parser.add_argument('--channel', '-c', default=0, type=int)
    parser.add_argument('--device', '-d', default='cuda:0', type=str)
    parser.add_argument('--overwrite', '-o', default=False, action='store_true')
    parser.add_argument('--upload', '-u', default=False, action='store_true')
    parser.add_argument('--username', default=None, type=str)
    parser.add_argument('--password', default=None, type=str)
    parser.add_argument('--host', default=None, type=str)
    parser.add_argument('--port', default=None, type=int)
    parser.parse_args()
    try:
        import omero  # noqa: F401
    except ImportError:
        raise SystemExit('omero-py is not installed: OMERO I/O is outside the MI355X hot path. '
                         'Export the images and run infer_script_local.py')
    raise SystemExit('OMERO transport is provided by the reference GUI stack; see INTEGRATION.md')


if __name__ == "__main__":
    main()

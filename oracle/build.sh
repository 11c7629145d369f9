#!/bin/bash
# Builds the CPU oracle of the post-processing (test infrastructure only) -> oracle/_build/liboracle_postproc.so
set -e
cd "$(dirname "$0")"
mkdir -p _build
gcc -O2 -ffp-contract=off -fPIC -shared -o _build/liboracle_postproc.so postproc_ref.c -lm
echo "built oracle/_build/liboracle_postproc.so"

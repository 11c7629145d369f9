#!/bin/bash
# GPU box: rocprofv3 kernel-trace stats of the bench command (summary -> gpurun_out/prof_<tag>/).
# usage: tools/prof_bench.sh <tag> [bench args...]
set -e
tag=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --train-only "$@" --no-cpu-baseline --no-kernel-timing --no-inference --no-bf16-block > $out/bench.log 2>&1
cd $out
f=$(find . -name '*kernel_stats.csv' | head -1)
echo "stats file: $f"
head -30 "$f"
# drop the bulky per-dispatch trace, keep the stats
find . -name '*kernel_trace.csv' -size +5M -delete || true

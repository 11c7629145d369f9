#!/bin/bash
# GPU box: rocprofv3 kernel stats of the inference block of bench.py (2048 x 2048 frames, network + post-processing).
# usage: tools/prof_infer.sh <tag> [bench args, e.g. --precision bf16]
set -e
tag=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_infer_$tag
mkdir -p $out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --inference-only --infer-frames 8 "$@" > $out/bench.log 2>&1
cd $out
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open("trace_kernel_stats.csv")))
for r in rows[:30]:
    print(r["Name"][:80].ljust(80), r["Calls"].rjust(5), "%9.1f us avg" % (float(r["AverageNs"]) / 1e3), "%8.2f ms total" % (float(r["TotalDurationNs"]) / 1e6))
PY
find . -name '*kernel_trace.csv' -size +5M -delete || true

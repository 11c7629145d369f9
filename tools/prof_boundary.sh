#!/bin/bash
# GPU box: rocprofv3 kernel stats of the boundary method's post-processing (tools/bench_boundary.py: synthetic 2048^2 frame,
# closed-form marker phase and heap replay, 4 calls each).
set -e
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_boundary_$1
mkdir -p $out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 $GRAFT_REPO_ROOT/tools/bench_boundary.py > $out/run.log 2>&1
grep -v "^W2\|^E2\|^I2" $out/run.log | tail -5
python3 - $out/t_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(4), "%10.1f us avg" % (float(r['AverageNs']) / 1e3), "%9.2f ms total" % (float(r['TotalDurationNs']) / 1e6))
PY
find $out -name '*kernel_trace.csv' -delete || true

#!/bin/bash
# GPU box: rocprofv3 kernel stats of the distance post-processing on the bench's synthetic 2048x2048 maps.
set -e
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_pp_$1
mkdir -p $out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 $GRAFT_REPO_ROOT/tools/bench_postproc.py 2048 2500 pp-only > $out/run.log 2>&1
cat $out/run.log | grep -v "^W2\|^E2\|^I2"
python3 - $out/t_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(r['Name'][:64].ljust(64), r['Calls'].rjust(4), "%9.1f us avg" % (float(r['AverageNs']) / 1e3), r['Percentage'])
PY
find $out -name '*kernel_trace.csv' -delete || true

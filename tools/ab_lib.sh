#!/bin/bash
# GPU box: the same bench command under several settings, alternating, on ONE box (boxes of the pool differ by up to 8 %,
# and one box drifts by 2-3 % between runs: compare medians of several rounds).  A setting is a build of the library
# (a file name in microbeseg_amd/, exported as MSEG_HIP_LIB) or an environment assignment VAR=value.
# usage: tools/ab_lib.sh "<setting1> <setting2> ..." [bench args...]
sets=$1; shift
for i in 1 2 3 4 5; do
for s in $sets; do
if [[ "$s" == *=* ]]; then pre="$s"; else pre="MSEG_HIP_LIB=$PWD/microbeseg_amd/$s"; fi
env $pre python bench.py --train-only --steps 30 --warmup 5 --no-inference --no-cpu-baseline --no-kernel-timing --no-bf16-block "$@" 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$s', d['value'], d['ms_per_step'], flush=True)
"
done; done | tee /tmp/ab.txt
python - <<'P'
import collections, statistics
r = collections.defaultdict(list)
for l in open('/tmp/ab.txt'):
    k, v, ms = l.split()
    r[k].append(float(ms))
for k, v in r.items():
    print(f"{k:28s} median {statistics.median(v):8.3f} ms  min {min(v):8.3f}  max {max(v):8.3f}")
P

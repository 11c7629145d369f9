#!/bin/bash
# GPU box: PMC counters of the MFMA kernels on the per-layer micro-benchmark (own run, no tracing flags besides --kernel-trace).
# usage: tools/pmc_kernels.sh <tag> <which: fwd|dgrad|wgrad|all>
set -e
tag=$1; which=${2:-all}
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT \
  --output-format csv -d $out/p1 -o t -- python3 $GRAFT_REPO_ROOT/tools/bench_kernels.py 32 $which > $out/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS \
  --output-format csv -d $out/p2 -o t -- python3 $GRAFT_REPO_ROOT/tools/bench_kernels.py 32 $which > $out/p2.log 2>&1
cd $out
python3 - <<'PY'
import csv, glob, collections
for p in ("p1", "p2"):
    f = glob.glob(f"{p}/**/*counter_collection.csv", recursive=True)
    if not f:
        print("no counter file for", p); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key); cnt[k] += 1
    for k in agg:
        if "igemm" in k or "wgrad_kernel" in k:
            print(p, k, "dispatches", cnt[k])
            for c, v in sorted(agg[k].items()):
                print(f"    {c:28s} {v:.4g}")
PY
find . -name '*.csv' -size +20M -delete || true

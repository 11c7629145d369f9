#!/bin/bash
# GPU box: per-kernel averages of a few hardware counters for one bench.py configuration, one rocprofv3 --pmc pass per
# counter group (no tracing domain besides --kernel-trace).  Output: gpurun_out/counters_<tag>.json + a table.
# usage: tools/pmc_counters.sh <tag> [bench args...]
set -e
tag=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_counters_$tag
mkdir -p $out
cd /tmp
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/g$i -o t -- python3 $GRAFT_REPO_ROOT/bench.py --train-only --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-inference "$@" > $out/g$i.log 2>&1 || echo "group $i failed: $grp"
done
cd $out
python3 - "$tag" <<'PY'
import csv, glob, collections, json, sys
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob("g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = k[5:] if k.startswith("void ") else k
        k = k.split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]].add(r["Dispatch_Id"])
res = {k: {c: v / max(len(cnt[k][c]), 1) for c, v in d.items()} for k, d in agg.items()}
json.dump(res, open(f"../counters_{tag}.json", "w"), indent=1, sort_keys=True)
for k, d in sorted(res.items()):
    if not any(s in k for s in ("igemm", "wgrad_halo", "norm_pass")):
        continue
    g = lambda c: d.get(c, float("nan"))
    hit = g("TCC_HIT_sum") / max(g("TCC_HIT_sum") + g("TCC_MISS_sum"), 1)
    print(f"{k[:46]:46s} mfma_busy/busy {g('SQ_VALU_MFMA_BUSY_CYCLES') / max(g('SQ_BUSY_CYCLES'), 1):6.3f}  L2 hit {hit:5.3f}  "
          f"L2 rd lat {g('TCP_TCC_READ_REQ_LATENCY_sum') / max(g('TCP_TCC_READ_REQ_sum'), 1):7.0f} clk  "
          f"lds conflict/active {g('SQ_LDS_BANK_CONFLICT') / max(g('SQ_LDS_IDX_ACTIVE'), 1):5.3f}  "
          f"wait_lds/wave_cycles {g('SQ_WAIT_INST_LDS') / max(g('SQ_WAVE_CYCLES'), 1):5.3f}  "
          f"wait_any/wave_cycles {g('SQ_WAIT_ANY') / max(g('SQ_WAVE_CYCLES'), 1):5.3f}")
PY

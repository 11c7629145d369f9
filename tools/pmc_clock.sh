#!/bin/bash
# GPU box: the clock the chip HOLDS under each kernel of a bench.py configuration (MI355X_MICROARCH.md, "DVFS give-back":
# effective clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel wall time; MFMA-dense bf16 loops on random data run well under the
# 2.4 GHz the spec peak is quoted at).  One rocprofv3 --pmc pass (kernel trace only, no other tracing domain).
# Output: gpurun_out/clock_<tag>.json + a table: per kernel the average duration, the effective clock, and the matrix-core
# busy share of the cycles that actually elapsed (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / elapsed cycles).
# usage: tools/pmc_clock.sh <tag> [bench args...]
set -e
tag=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_clock_$tag
mkdir -p $out
cd /tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $out/g1 -o t -- python3 $GRAFT_REPO_ROOT/bench.py --train-only --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-inference --no-bf16-block "$@" > $out/g1.log 2>&1 || echo "pass failed"
cd $out
python3 - "$tag" <<'PY'
import csv, glob, collections, json, sys
tag = sys.argv[1]
dur = {}
for f in glob.glob("g*/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
per = collections.defaultdict(lambda: collections.defaultdict(dict))
for f in glob.glob("g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = k[5:] if k.startswith("void ") else k
        k = k.split("(")[0]
        d = r["Dispatch_Id"]
        per[k][d][r["Counter_Name"]] = per[k][d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        if d not in dur and "Start_Timestamp" in r:
            dur[d] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
res = {}
for k, ds in per.items():
    ns = gui = mf = 0.0
    n = 0
    for d, c in ds.items():
        if d not in dur or "GRBM_GUI_ACTIVE" not in c:
            continue
        ns += dur[d]; gui += c["GRBM_GUI_ACTIVE"]; mf += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0); n += 1
    if n == 0 or ns <= 0:
        continue
    cyc = gui / 8.0                                   # cycles elapsed per XCD, summed over the launches
    res[k] = {"launches": n, "avg_us": ns / n / 1e3, "clock_ghz": cyc / ns,
              "mfma_busy_of_elapsed": mf / 1024.0 / cyc if cyc > 0 else None}
json.dump(res, open(f"../clock_{tag}.json", "w"), indent=1, sort_keys=True)
for k, d in sorted(res.items(), key=lambda kv: -kv[1]["avg_us"] * kv[1]["launches"]):
    if d["avg_us"] < 30:
        continue
    print(f"{k[:60]:60s} n {d['launches']:4d}  {d['avg_us']:8.1f} us  clock {d['clock_ghz']:5.2f} GHz  "
          f"mfma busy / elapsed {d['mfma_busy_of_elapsed']:5.3f}")
PY

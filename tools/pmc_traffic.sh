#!/bin/bash
# GPU box: HBM traffic of every kernel of one bench.py training step from the TCC counters, collected in two separate
# --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; no tracing domains besides --kernel-trace).
# Corrections per /opt/skills/guides/MI355X_MICROARCH.md §HBM: counter unit = KiB; on gfx950 FETCH_SIZE reports half
# of a wide coalesced read stream -> doubled.  Output: gpurun_out/traffic_<tag>.json (avg bytes per launch per kernel).
set -e
tag=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_traffic_$tag
mkdir -p $out
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$c -o t -- python3 $GRAFT_REPO_ROOT/bench.py --train-only --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-inference --no-bf16-block "$@" > $out/$c.log 2>&1
done
cd $out
python3 - "$tag" "$GRAFT_REPO_ROOT" "$@" <<'PY'
import csv, glob, collections, json, sys
tag, root = sys.argv[1], sys.argv[2]
sys.path.insert(0, root)
sys.argv = ["bench.py"] + sys.argv[3:]
import bench
config = bench.workload_key(bench.parse())       # bench.py picks the profile of ITS workload by this record ...
config["csrc_sha"] = bench.csrc_digest()         # ... and of ITS kernel sources (a profile of other sources is stale)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{c}/**/*counter_collection.csv", recursive=True)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != c:
            continue
        k = r["Kernel_Name"]
        k = k[5:] if k.startswith("void ") else k
        k = k.split("(")[0]
        agg[k][c] += float(r["Counter_Value"])
        cnt[(k, c)].add(r["Dispatch_Id"])
res = {}
for k, v in agg.items():
    n = max(len(cnt[(k, "FETCH_SIZE")]), 1)
    fetch = 2.0 * v["FETCH_SIZE"] * 1024 / n           # gfx950: FETCH_SIZE counts 64 B per 128 B request
    write = v["WRITE_SIZE"] * 1024 / max(len(cnt[(k, "WRITE_SIZE")]), 1)
    res[k] = {"launches": n, "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
              "hbm_bytes_per_launch": fetch + write}
res["__config__"] = config
json.dump(res, open(f"../traffic_{tag}.json", "w"), indent=1, sort_keys=True)
del res["__config__"]
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:12]:
    print(f"{k[:60]:60s} n={v['launches']:4d} fetch {v['fetch_bytes_per_launch']/1e6:9.1f} MB write {v['write_bytes_per_launch']/1e6:9.1f} MB")
PY
rm -rf FETCH_SIZE WRITE_SIZE

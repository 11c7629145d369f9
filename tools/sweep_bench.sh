#!/bin/bash
# GPU box: bench.py over the secondary configurations quoted in README.md / DESIGN.md (value + ms/step per line).
run() { printf "%-46s" "$*"; python bench.py --no-cpu-baseline --no-inference --no-bf16-block --no-kernel-timing --steps 8 --warmup 3 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'crops/s', d['ms_per_step'], 'ms/step')"; }
run
run --act mish --optimizer ranger
run --arch U
run --norm gn
run --norm in
run --size 320
run --size 320 --norm gn
run --filters 32 512
run --size 512 --batch 8
run --batch 2
run --batch 4
run --batch 8
run --batch 128
run --precision bf16
run --precision bf16 --size 320
run --precision bf16 --size 320 --norm gn
run --precision bf16 --size 320 --act mish --optimizer ranger
run --precision bf16 --arch U
run --precision bf16 --batch 4

#!/bin/bash
# Register / scratch / occupancy summary of every kernel in one HIP source (device-only compile, nothing is written in-tree)
# usage: tools/kernel_resources.sh microbeseg_amd/csrc/igemm.hip [filter-regex]
src=$1; filt=${2:-.}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -pragma-unroll-threshold=200000 -c "$src" --cuda-device-only -Rpass-analysis=kernel-resource-usage -o /dev/null 2>&1 \
 | grep -E "Function Name|VGPRs:|AGPRs:|ScratchSize|Occupancy|LDS Size" | sed 's/.*remark: [^ ]* *//; s/ \[-Rpass.*//' | paste - - - - - - | c++filt | grep -E "$filt"

#!/bin/bash
# dev tool: VGPR / AGPR / occupancy / spill table of every kernel in one .hip file (default igemm.hip).
# A change in the epilogue once moved a tile config from 6 to 2 waves per SIMD without any test noticing: run this
# after touching a kernel and compare.
f=${1:-igemm.hip}
cd "$(dirname "$0")/../yolo_dual_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I../../include -c "$f" -o /tmp/_kr.o \
    -Rpass-analysis=kernel-resource-usage 2>&1 |
    grep -E "Function Name|VGPRs:|AGPRs|Occupancy|VGPRs Spill" | sed 's/.*remark: //; s/\[-Rpass.*//' | paste - - - - -

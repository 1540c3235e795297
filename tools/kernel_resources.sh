#!/bin/bash
# dev tool: VGPR / AGPR / occupancy / spill table of every kernel in one .hip file (default igemm.hip).
# A change in the epilogue once moved a tile config from 6 to 2 waves per SIMD without any test noticing: run this
# after touching a kernel and compare.
#   tools/kernel_resources.sh [file.hip]            print the table
#   tools/kernel_resources.sh --check               igemm.hip: compare the step's hot instantiations with the floor table below
#                                                   (exit 1 and one line per violation)
check=0
if [ "$1" = "--check" ]; then check=1; shift; fi
f=${1:-igemm.hip}
cd "$(dirname "$0")/../yolo_dual_amd/csrc" || exit 1
tab=$(/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -Wno-unused-result -I../../include -I. -c "$f" -o /tmp/_kr.o \
    -Rpass-analysis=kernel-resource-usage 2>&1 |
    grep -E "Function Name|VGPRs:|AGPRs|Occupancy|VGPRs Spill" | sed 's/.*remark: //; s/\[-Rpass.*//' | paste - - - - -)
if [ $check = 0 ]; then echo "$tab"; exit 0; fi

# guarded instantiations (the ones the default bench step launches): mangled-name pattern, least waves/SIMD, most spilled VGPRs.
# A spill entry above 0 is a spill OUTSIDE the main loop that was looked at in the assembly (igemm2s<..,2,..>: one 8-byte store before
# the K loop and its reload after it).  The table is compiled with the flags of yolo_dual_amd/build.py.
guard='
igemm2_kernelILi128ELi128ELi8ELi4ELi2ELb0ELi0E 4 0
igemm2_kernelILi256ELi128ELi8ELi4ELi3ELb0ELi1E 2 0
igemm2_kernelILi256ELi64ELi8ELi8ELi3ELb0ELi0E 4 0
igemm2w_kernelILi64ELb1ELi4E 2 0
igemm2w_kernelILi64ELb0ELi4E 2 0
igemm2w_kernelILi128ELb1ELi4E 2 0
igemm2w_kernelILi128ELb0ELi4E 2 0
igemm2s_kernelILi8ELi4ELi2ELb0E 4 2
igemm2s_kernelILi8ELi4ELi2ELb1E 4 2
igemm2p_kernelILi128ELi128ELi8ELi4ELb0E 4 0
igemm2p_kernelILi128ELi128ELi8ELi4ELb1E 4 0
igemm2l_kernelILi256ELi128ELi8ELi4ELi3ELi4E 3 0
igemm2l_kernelILi128ELi128ELi8ELi4ELi3ELi4E 3 0
igemm2l_kernelILi128ELi128ELi4ELi2ELi2ELi4E 4 0
wgrad3s_kernelILi128ELi32ELi4ELi4E 4 0
wgrad3s_kernelILi64ELi32ELi4ELi4E 4 0
pwbw_kernelILi5ELb0E 5 0
pwbw_kernelILi4ELb1E 5 0
wgrad3_kernelILi128E 3 0
'
bad=0
while read -r pat occ spill; do
    [ -z "$pat" ] && continue
    rows=$(echo "$tab" | grep -F "$pat")
    if [ -z "$rows" ]; then echo "MISSING  $pat"; bad=1; continue; fi
    while IFS= read -r row; do
        o=$(echo "$row" | sed -n 's/.*Occupancy \[waves\/SIMD\]: \([0-9]*\).*/\1/p')
        s=$(echo "$row" | sed -n 's/.*VGPRs Spill: \([0-9]*\).*/\1/p')
        if [ "$o" -lt "$occ" ] || [ "$s" -gt "$spill" ]; then echo "REGRESSED  $pat: occupancy $o (floor $occ), spilled $s (ceiling $spill)"; bad=1; fi
    done <<< "$rows"
done <<< "$guard"
[ $bad = 0 ] && echo "kernel resources: all guarded instantiations at or above their floors"
exit $bad

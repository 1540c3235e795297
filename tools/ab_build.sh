#!/bin/bash
# dev tool: build a variant of libydl_hip.so with extra -D flags on igemm.hip (other objects are reused from the main build) for
# same-box A/B runs:   tools/ab_build.sh prio1 -DYDL_PRIO=1   ->  yolo_dual_amd/lib/ab_prio1.so   (select with YDL_LIB=...)
set -e
name=$1; shift
cd "$(dirname "$0")/../yolo_dual_amd"
src=${AB_SRC:-igemm}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -Wno-unused-result -I../include -Icsrc "$@" -c csrc/$src.hip -o lib/ab_${name}_$src.o
objs=""
for o in err replay igemm bn spatial loss optim dcnv3 dcn_blocks input; do
  if [ "$o" = "$src" ]; then objs="$objs lib/ab_${name}_$src.o"; else objs="$objs lib/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/ab_${name}.so $objs
echo "built yolo_dual_amd/lib/ab_${name}.so"

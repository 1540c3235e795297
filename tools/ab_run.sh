#!/bin/bash
# dev tool (GPU box): the same conv_bench command under several builds of the library, one after the other on one box.
# usage: tools/ab_run.sh "base new prio1" fwd --only 1,5,9 ...      (name "new" = the main build, others = lib/ab_<name>.so)
libs=$1; shift
cd "$(dirname "$0")/.."
for l in $libs; do
  if [ "$l" = "new" ]; then unset YDL_LIB; else export YDL_LIB=$PWD/yolo_dual_amd/lib/ab_$l.so; fi
  echo "=== $l"
  python tools/conv_bench.py "$@" 2>&1 | grep -v amdgpu.ids
done

#!/bin/bash
# usage: tools/experiments/build_variant.sh NAME "-DPT_WF_FLAT=0 ..."   -> tools/experiments/variants/NAME.so
# Rebuilds only the translation units the -D flags can change (pt_wavefront.hip, pt_kernels.hip) and links them with the
# default objects; tools/experiments/variants.py then benches every variant through PT_PLUGIN.
set -e
NAME=$1; shift
DEFS="$*"
cd "$(dirname "$0")/../../unity_webgpu_pathtracer_amd/csrc"
make -s >/dev/null
mkdir -p ../../tools/experiments/variants/_obj_$NAME
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fvisibility=hidden -I../../include -Wno-unused-function -Wno-unused-value -mllvm -amdgpu-sched-strategy=max-memory-clause $EXTRA_FLAGS"
for f in pt_wavefront pt_kernels pt_api; do
  hipcc --offload-arch=gfx950 $FLAGS $DEFS -c $f.hip -o ../../tools/experiments/variants/_obj_$NAME/$f.o &
done
wait
OBJS=""
for o in $(ls _obj/*.o | grep -v "/stress_"); do
  b=$(basename $o .o)
  if [ -f ../../tools/experiments/variants/_obj_$NAME/$b.o ]; then OBJS="$OBJS ../../tools/experiments/variants/_obj_$NAME/$b.o"; else OBJS="$OBJS $o"; fi
done
hipcc --offload-arch=gfx950 -shared -o ../../tools/experiments/variants/$NAME.so $OBJS -ldl 2>&1 | tail -3
echo built tools/experiments/variants/$NAME.so

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
# STRATEGY=name replaces the machine-scheduler strategy of the default build (max-memory-clause), STRATEGY=none drops the flag.
# BASE_EXTRA: what csrc/Makefile adds to every device translation unit (default: its -mllvm -amdgpu-use-amdgpu-trackers=1);
# A_EXTRA: what it adds to the second compilation of pt_wavefront.hip, the default schedule's kernels (default: -mllvm -enable-post-misched=false)
STRAT="-mllvm -amdgpu-sched-strategy=${STRATEGY:-max-memory-clause}"
[ "$STRATEGY" = "none" ] && STRAT=""
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fvisibility=hidden -I../../include -Wno-unused-function -Wno-unused-value $STRAT ${BASE_EXTRA--mllvm -amdgpu-use-amdgpu-trackers=1} $EXTRA_FLAGS"
OUT=../../tools/experiments/variants/_obj_$NAME
for f in pt_kernels pt_api; do
  hipcc --offload-arch=gfx950 $FLAGS $DEFS -c $f.hip -o $OUT/$f.o &
done
hipcc --offload-arch=gfx950 $FLAGS $DEFS -DPT_WF_TU_B -c pt_wavefront.hip -o $OUT/pt_wavefront.o &
hipcc --offload-arch=gfx950 $FLAGS ${A_EXTRA--mllvm -enable-post-misched=false} $DEFS -c pt_wavefront.hip -o $OUT/pt_wavefront_a.o &
wait
OBJS=""
for o in $(ls _obj/*.o | grep -v "/stress_"); do
  b=$(basename $o .o)
  if [ -f ../../tools/experiments/variants/_obj_$NAME/$b.o ]; then OBJS="$OBJS ../../tools/experiments/variants/_obj_$NAME/$b.o"; else OBJS="$OBJS $o"; fi
done
hipcc --offload-arch=gfx950 -shared -o ../../tools/experiments/variants/$NAME.so $OBJS -ldl 2>&1 | tail -3
echo built tools/experiments/variants/$NAME.so

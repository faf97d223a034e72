#!/bin/bash
run() { python3 bench.py $ARGS --no-extra --no-cpu-baseline --no-alg-replay > gpurun_out/small_tmp.json 2>/dev/null; python3 -c "import json; j=json.load(open('gpurun_out/small_tmp.json')); print('$1', j['value'], j['ms_per_step'])"; }
ARGS="--steps 24 --warmup 6"; run warm > /dev/null
for q in 8 16; do
  export GPU_MAX_HW_QUEUES=$q
  ARGS="--steps 36 --warmup 12"
  unset PT_PLUGIN; run "1080p sets6 q$q"
  for v in sets8 sets12; do export PT_PLUGIN=$PWD/tools/experiments/variants/$v.so; run "1080p $v q$q"; done
done

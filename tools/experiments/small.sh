#!/bin/bash
# usage (GPU box): bash tools/experiments/small.sh -- 960x540 / 8 spp with more state sets / hardware queues
run() { python3 bench.py --width 960 --height 540 --steps 60 --warmup 12 --no-extra --no-cpu-baseline --no-alg-replay > gpurun_out/small_tmp.json 2>/dev/null; python3 -c "import json; j=json.load(open('gpurun_out/small_tmp.json')); print('$1', j['value'], j['ms_per_step'])"; }
run warm > /dev/null
for q in 8 16 24; do
  export GPU_MAX_HW_QUEUES=$q
  unset PT_PLUGIN; run "sets6 q$q"
  for v in sets8 sets12; do export PT_PLUGIN=$PWD/tools/experiments/variants/$v.so; run "$v q$q"; done
done

#!/bin/bash
# usage (GPU box): bash tools/experiments/iters.sh  -- Mrays/s for different numbers of wavefront iterations before the cleanup kernel
python3 bench.py --no-extra --no-cpu-baseline --no-alg-replay > /dev/null 2>&1
for n in 0 44 40 36 60; do
  python3 bench.py --steps 24 --warmup 5 --no-extra --no-cpu-baseline --no-alg-replay --wf-iterations $n > gpurun_out/iters_$n.json 2>/dev/null
  python3 -c "import json; j=json.load(open('gpurun_out/iters_$n.json')); print($n, j['value'], j['ms_per_step'])"
done

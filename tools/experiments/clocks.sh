#!/bin/bash
# usage (GPU box): bash tools/experiments/clocks.sh  -> shader clock / power sampled every 0.5 s while bench.py's main workload runs (60 steps)
( for i in $(seq 1 24); do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Power (W)\|Socket" | tr '\n' ' '; echo; sleep 0.5; done ) > gpurun_out/clocks_samples.log &
SAMPLER=$!
sleep 1
python bench.py --steps 300 --warmup 5 --no-extra --no-cpu-baseline --no-alg-replay 2>/dev/null | grep "^{" | python -c 'import sys, json; j = json.loads(sys.stdin.read()); print("Mrays/s", j["value"], "ms", j["ms_per_step"])'
wait $SAMPLER
cat gpurun_out/clocks_samples.log

#!/bin/bash
# schedule 1, ONE pass in flight: kernel-time sums per pass of the trace / shade kernels at 4 vs 8 trace waves per SIMD
export TMPDIR=/tmp
for v in tr8 tr4; do
  PT_PLUGIN=$PWD/tools/experiments/variants/$v.so timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/occ_${v}_trace -- python3 bench.py --schedule 1 --passes-in-flight 1 --steps 3 --warmup 1 --latency-steps 0 --no-extra --no-cpu-baseline --no-alg-replay > gpurun_out/occ_$v.log 2>&1
  echo "== $v"; python3 tools/experiments/prof4_sum.py occ_$v 4 | grep trace
done

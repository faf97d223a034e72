#!/bin/bash
export TMPDIR=/tmp
CMD="python3 bench.py --schedule 1 --passes-in-flight 1 --steps 3 --warmup 1 --latency-steps 0 --no-extra --no-cpu-baseline --no-alg-replay"
for v in 0 1; do
  PT_WF_SHADE_SORTED=$v timeout -k 10 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d gpurun_out/srt${v}_valu -- $CMD > gpurun_out/srt${v}.log 2>&1
  PT_WF_SHADE_SORTED=$v timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/srt${v}_trace -- $CMD > gpurun_out/srt${v}t.log 2>&1
  echo "== sorted $v"; python3 tools/experiments/prof4_sum.py srt$v 4 | grep -E "shade|trace_refill"
done
echo "== 4 waves"; PT_PLUGIN=$PWD/tools/experiments/variants/sh4.so bash tools/experiments/sorted.sh | tail -1

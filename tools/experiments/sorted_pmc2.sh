#!/bin/bash
export TMPDIR=/tmp
CMD="python3 bench.py --schedule 1 --passes-in-flight 1 --steps 3 --warmup 1 --latency-steps 0 --no-extra --no-cpu-baseline --no-alg-replay"
for v in 0 2; do
  PT_WF_SHADE_SORTED=$v timeout -k 10 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d gpurun_out/sp${v}_valu -- $CMD > gpurun_out/sp${v}.log 2>&1
  PT_WF_SHADE_SORTED=$v timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/sp${v}_trace -- $CMD > gpurun_out/sp${v}t.log 2>&1
  echo "== shade mode $v"; python3 tools/experiments/prof4_sum.py sp$v 4 | grep -E "shade"
done

#!/bin/bash
# usage (GPU box): bash tools/experiments/pmc_icache.sh TAG  -- instruction-cache and issue-side counters per kernel
TAG=$1
export TMPDIR=/tmp
CMD="python3 bench.py --steps 4 --warmup 1 --no-extra --no-cpu-baseline --no-alg-replay $BENCH_EXTRA"
i=0
while read -r GROUP; do
  [ -z "$GROUP" ] && continue
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $GROUP --kernel-trace --output-format csv -d gpurun_out/${TAG}_i$i -- $CMD > gpurun_out/${TAG}_i$i.log 2>&1 || { echo "group $i failed: $GROUP"; tail -3 gpurun_out/${TAG}_i$i.log; continue; }
  python3 tools/experiments/pmc_sum.py gpurun_out/${TAG}_i$i
done <<'GROUPS'
SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
SQ_IFETCH SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU
SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT
SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM
GROUPS

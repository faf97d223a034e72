#!/bin/bash
# usage (on the GPU box): bash tools/experiments/pmc_groups.sh TAG [variant.so]
# One rocprofv3 --pmc run per counter group over the short bench command; sums per kernel printed by pmc_sum.py.
TAG=$1
[ -n "$2" ] && export PT_PLUGIN=$(realpath $2)
export TMPDIR=/tmp
CMD="python3 bench.py --steps 4 --warmup 1 --no-extra --no-cpu-baseline --no-alg-replay $BENCH_EXTRA"
i=0
while read -r GROUP; do
  [ -z "$GROUP" ] && continue
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $GROUP --kernel-trace --output-format csv -d gpurun_out/${TAG}_g$i -- $CMD > gpurun_out/${TAG}_g$i.log 2>&1 || { echo "group $i failed: $GROUP"; tail -3 gpurun_out/${TAG}_g$i.log; continue; }
  python3 tools/experiments/pmc_sum.py gpurun_out/${TAG}_g$i
done <<'GROUPS'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD
TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
TD_TD_BUSY_sum TD_TC_STALL_sum
GRBM_GUI_ACTIVE
SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH
GROUPS

#!/bin/bash
# usage (GPU box): bash tools/experiments/tlas_pmc.sh TAG [variant.so]  -> per-kernel counter sums of 4 passes of the HAS_TLAS scene
TAG=$1
[ -n "$2" ] && export PT_PLUGIN=$(realpath $2)
export TMPDIR=/tmp
i=0
while read -r GROUP; do
  [ -z "$GROUP" ] && continue
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $GROUP --kernel-trace --output-format csv -d gpurun_out/${TAG}_g$i -- python3 tools/experiments/tlas_run.py 4 1 > gpurun_out/${TAG}_g$i.log 2>&1 || { echo "group $i failed: $GROUP"; tail -3 gpurun_out/${TAG}_g$i.log; continue; }
  python3 tools/experiments/pmc_sum.py gpurun_out/${TAG}_g$i
done <<'GROUPS'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD
SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY
TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE
GROUPS

#!/bin/bash
# Counter passes over the fused persistent kernel (schedule 4), one pass in flight.  usage: bash tools/experiments/prof4.sh <tag> [schedule]
TAG=${1:-p4}; SCHED=${2:-4}
export TMPDIR=/tmp
CMD="python3 bench.py --schedule $SCHED --passes-in-flight 1 --steps 3 --warmup 1 --latency-steps 0 --no-extra --no-cpu-baseline --no-alg-replay"
run() { name=$1; shift; timeout -k 10 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/${TAG}_$name -- $CMD > gpurun_out/${TAG}_$name.log 2>&1 || echo "pass $name failed"; }
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_trace -- $CMD > gpurun_out/${TAG}_trace.log 2>&1
run valu SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run wait SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU
run lane SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 tools/experiments/prof4_sum.py $TAG

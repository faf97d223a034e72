#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root:  bash profiles/collect.sh r03
# Separate rocprofv3 runs of the SAME command bench.py's default run times (--steps 20 --warmup 5, passes pipelined on the library's
# default number of state sets): kernel trace + stats, then one --pmc pass per counter group (FETCH_SIZE and WRITE_SIZE do not fit one
# pass; --pmc is never combined with other trace domains; the program itself follows `--`).  A second kernel trace with ONE pass
# in flight (--passes-in-flight 1: the kernels of a pass run back to back, nothing overlaps them) gives honest per-kernel
# durations.  profiles/summarize.py turns the CSVs into the summaries kept under profiles/ and says which regime each number is from.
set -e
TAG=${1:-r03}
export TMPDIR=/tmp
CMD="python3 bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-alg-replay --latency-steps 0"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_trace -- $CMD > gpurun_out/${TAG}_trace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_serial -- $CMD --passes-in-flight 1 > gpurun_out/${TAG}_serial.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${TAG}_fetch -- $CMD > gpurun_out/${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/${TAG}_write -- $CMD > gpurun_out/${TAG}_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/${TAG}_valu -- $CMD > gpurun_out/${TAG}_valu.log 2>&1
# where the wave-cycles go: WAIT_ANY (parked on s_waitcnt) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY ~ WAVE_CYCLES (quad-cycles)
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --kernel-trace --output-format csv -d gpurun_out/${TAG}_wait -- $CMD > gpurun_out/${TAG}_wait.log 2>&1
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d gpurun_out/${TAG}_lane -- $CMD > gpurun_out/${TAG}_lane.log 2>&1
# the vector-memory path and the dispatcher (one block's counters per pass: TA and TD hold two each, rocprofv3 aborts on more)
for G in "ta:TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" "td:TD_TD_BUSY_sum TD_TC_STALL_sum" \
         "tcp:TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
         "tcc:TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "grbm:GRBM_GUI_ACTIVE"; do
  NAME=${G%%:*}; COUNTERS=${G#*:}
  timeout -k 10 200 rocprofv3 --pmc $COUNTERS --kernel-trace --output-format csv -d gpurun_out/${TAG}_$NAME -- $CMD > gpurun_out/${TAG}_$NAME.log 2>&1 || echo "pass $NAME failed"
done
tail -1 gpurun_out/${TAG}_trace.log | cut -c1-600

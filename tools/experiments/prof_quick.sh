#!/bin/bash
# usage (GPU box): bash tools/experiments/prof_quick.sh TAG [variant.so]  -- kernel trace (concurrent) + one serialized SQ counter pass
TAG=$1
[ -n "$2" ] && export PT_PLUGIN=$(realpath $2)
export TMPDIR=/tmp
CMD="python3 bench.py --steps 4 --warmup 1 --no-extra --no-cpu-baseline --no-alg-replay $BENCH_EXTRA"
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_kt -- $CMD > gpurun_out/${TAG}_kt.log 2>&1
python3 tools/experiments/pmc_sum.py gpurun_out/${TAG}_kt
python3 - "$TAG" <<'PY'
import csv, glob, sys
f = glob.glob(f"gpurun_out/{sys.argv[1]}_kt/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "pt_wf" in r["Kernel_Name"]]
ev = []
for r in rows: ev += [(int(r["Start_Timestamp"]), 1), (int(r["End_Timestamp"]), -1)]
ev.sort()
t0, t1 = ev[0][0], ev[-1][0]
busy = [0.0] * 8; cur = 0; last = t0
for t, d in ev:
    busy[min(cur, 7)] += t - last; last = t; cur += d
span = t1 - t0
print("span ms", span / 1e6, "fraction of time with k kernels running:", [round(b / span, 3) for b in busy[:5]])
PY
timeout -k 10 150 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d gpurun_out/${TAG}_sq -- $CMD > gpurun_out/${TAG}_sq.log 2>&1
python3 tools/experiments/pmc_sum.py gpurun_out/${TAG}_sq

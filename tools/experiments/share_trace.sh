#!/bin/bash
# usage (GPU box): bash tools/experiments/share_trace.sh WORLD -- kernel trace of rank 0's share of a WORLD-way sharded frame
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=16
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/share_kt -- python3 tools/experiments/share8.py $1 > gpurun_out/share_kt.log 2>&1
tail -2 gpurun_out/share_kt.log | head -1
python3 tools/experiments/overlap.py gpurun_out/share_kt
python3 - <<'PY'
import csv, glob, collections
f = sorted(glob.glob("gpurun_out/share_kt/*/*kernel_trace.csv"))[-1]
rows = [r for r in csv.DictReader(open(f)) if "pt_wf" in r["Kernel_Name"]]
d = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    k = "shade" if "shade" in n else "tail" if ("refill" in n and "true" in n.split("refill")[1][:14]) else "main" if "refill" in n else "other"
    d[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    v.sort()
    print(k, "calls", len(v), "mean us", round(sum(v) / len(v), 1), "median", round(v[len(v) // 2], 1), "p10", round(v[len(v) // 10], 1), "p90", round(v[len(v) * 9 // 10], 1))
PY

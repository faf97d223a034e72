#!/usr/bin/env python3
"""Sums the counters of tools/experiments/prof4.sh per kernel and per pass (4 passes = steps 3 + warmup 1)."""
import csv, glob, os, sys, re, collections
tag = sys.argv[1]; passes = int(sys.argv[2]) if len(sys.argv) > 2 else 4
base = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "gpurun_out")
def short(n):
    m = re.search(r"(pt_\w+)", n); return m.group(1) if m else n[:30]
for name in ("trace", "valu", "wait", "lane", "fetch", "write"):
    files = sorted(glob.glob(os.path.join(base, f"{tag}_{name}", "*", "*_counter_collection.csv")), key=os.path.getmtime)
    if name == "trace":
        tf = sorted(glob.glob(os.path.join(base, f"{tag}_trace", "*", "*_kernel_trace.csv")), key=os.path.getmtime)
        if not tf: continue
        d = collections.defaultdict(list)
        for r in csv.DictReader(open(tf[-1])):
            if "pt_" in r["Kernel_Name"]: d[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
        for k, v in d.items(): print(f"[trace] {k:28s} calls {len(v):4d}  mean {sum(v)/len(v):9.3f} ms  total/pass {sum(v)/passes:9.3f} ms")
        continue
    if not files: continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(files[-1])):
        if "pt_" not in r["Kernel_Name"]: continue
        acc[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, c in acc.items():
        print(f"[{name}] {k:24s} " + "  ".join(f"{cn}={v/passes:.4g}" for cn, v in sorted(c.items())))

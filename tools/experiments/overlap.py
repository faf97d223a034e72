"""Concurrency of the render kernels in steady state: fraction of time with k kernels running, from a rocprofv3 kernel trace
(usage: overlap.py <dir with *_kernel_trace.csv>).  The first and last 20 % of the span (pipeline fill / drain) are cut."""
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[-1]
rows = [r for r in csv.DictReader(open(f)) if "pt_wf" in r["Kernel_Name"]]
t0 = min(int(r["Start_Timestamp"]) for r in rows); t1 = max(int(r["End_Timestamp"]) for r in rows)
lo, hi = t0 + (t1 - t0) * 0.2, t0 + (t1 - t0) * 0.8
ev = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    kind = "shade" if "shade" in r["Kernel_Name"] else ("tail" if "true" in r["Kernel_Name"].split("refill")[-1][:14] else "main") if "refill" in r["Kernel_Name"] else "other"
    ev += [(s, 1, kind), (e, -1, kind)]
ev.sort()
cur = 0; last = t0; busy = collections.Counter(); kinds = collections.Counter(); mix = collections.Counter()
for t, d, k in ev:
    a, b = max(last, lo), min(t, hi)
    if b > a:
        busy[min(cur, 6)] += b - a
        mix[tuple(sorted((kk, v) for kk, v in kinds.items() if v))] += b - a
    last = t; cur += d; kinds[k] += d
span = hi - lo
print("steady-state fraction of time with k kernels running:", {k: round(v / span, 3) for k, v in sorted(busy.items())})
print("most common mixes:")
for m, v in mix.most_common(8): print("  ", round(v / span, 3), dict(m))

#!/usr/bin/env python3
"""Turns the rocprofv3 CSVs under gpurun_out/<tag>_{trace,fetch,write}/ into the small summaries kept in profiles/.

usage: python profiles/summarize.py <tag> [workload]     e.g.  python profiles/summarize.py r01 sponza
writes profiles/<tag>_kernel_stats.csv (copy of rocprofv3 --stats kernel summary), profiles/<tag>_pmc.json and
profiles/hbm_traffic.json (bytes per launch of the dominant kernel, read by bench.py for roofline.traffic).

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are collected in SEPARATE
--pmc passes (TCC slots), both are in KiB units -> x1024; on gfx950 FETCH_SIZE under-reports wide coalesced streaming
reads by exactly 2x, so the fetch side is reported both raw and doubled; this kernel's reads are scattered 16-byte
gathers for which the counter is uncalibrated, so the doubled figure is an upper estimate."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1]
    workload = sys.argv[2] if len(sys.argv) > 2 else "sponza"
    base = os.path.join(ROOT, "gpurun_out")
    stats = glob.glob(os.path.join(base, f"{tag}_trace", "*", "*_kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"))
    trace = glob.glob(os.path.join(base, f"{tag}_trace", "*", "*_kernel_trace.csv"))
    out = {"tag": tag, "workload": workload}
    if trace:
        rows = [r for r in csv.DictReader(open(trace[0])) if "pt_" in r["Kernel_Name"]]
        per = {}
        for r in rows:
            per.setdefault(r["Kernel_Name"].split("(")[0], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
        out["kernel_ms"] = {k: {"calls": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)} for k, v in per.items()}
        r0 = rows[0]
        out["resources"] = {k: r0.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size_X", "Grid_Size_X")}
    pmc = {}
    for kind, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        f = glob.glob(os.path.join(base, f"{tag}_{kind}", "*", "*_counter_collection.csv"))
        if not f:
            continue
        vals = {}
        for r in csv.DictReader(open(f[0])):
            if "pt_" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals.setdefault(r["Kernel_Name"].split("(")[0], []).append(float(r["Counter_Value"]))
        pmc[counter] = {k: {"launches": len(v), "mean_KiB": sum(v) / len(v)} for k, v in vals.items()}
    out["pmc"] = pmc
    dom = None
    if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
        dom = max(pmc["FETCH_SIZE"], key=lambda k: pmc["FETCH_SIZE"][k]["mean_KiB"])
        fetch = pmc["FETCH_SIZE"][dom]["mean_KiB"] * 1024
        write = pmc["WRITE_SIZE"].get(dom, {"mean_KiB": 0})["mean_KiB"] * 1024
        out["hbm_bytes_per_launch"] = {"kernel": dom, "fetch_raw": fetch, "fetch_x2_upper": 2 * fetch, "write": write,
                                       "total_raw": fetch + write, "total_upper": 2 * fetch + write}
        json.dump({"workload": workload, "n_gpus": 1, "kernel": dom, "bytes_per_launch": round(2 * fetch + write),
                   "note": "FETCH_SIZE x2 (gfx950 correction, upper estimate for scattered 16-B reads) + WRITE_SIZE, separate --pmc passes; source profiles/%s_pmc.json" % tag},
                  open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"), indent=1)
    json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_pmc.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

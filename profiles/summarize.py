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


def short(name):
    import re
    m = re.search(r"(pt_\w+)(<[\w, ]+>)?", name)
    return (m.group(1) + (m.group(2) or "").replace(" ", "")) if m else name[:40]


def measured_valu_peak():
    """Wave64 VALU issue peak of this chip, MEASURED by tools/microbench/valu_issue.hip (independent v_fma_f32 streams, 8 waves
    per SIMD, every CU): profiles/r02_valu_issue.json.  Round 1 assumed 4 cycles per wave instruction (6.1e11/s); the measurement
    says 2.3 cycles at the nominal clock = 1.08e12/s (the microarchitecture guide: SIMD-32, 2 cycles)."""
    f = os.path.join(ROOT, "profiles", "r02_valu_issue.json")
    rec = json.load(open(f))
    best = max(r["wave_insts_per_s"] for r in rec["results"] if r["kind"] == "indep")
    return best, "profiles/r02_valu_issue.json (tools/microbench/valu_issue.hip: independent v_fma_f32, 8 waves/SIMD, all CUs)"


def per_pass(rows, value_of):
    """Sum a per-dispatch quantity over the kernels of each pass.  A pass is one pt_megakernel dispatch, or the
    pt_wf_init ... pt_wf_fold_rows sequence of the wavefront schedule (passes overlap in time but every dispatch
    belongs to exactly one pass; with overlapping passes dispatches are attributed by their order per kernel name)."""
    totals, per_kernel = [], {}
    mk = [value_of(r) for r in rows if "pt_megakernel" in r["Kernel_Name"]]
    if mk:
        return mk, {"pt_megakernel": sum(mk) / len(mk)}
    n_pass = sum(1 for r in rows if "pt_wf_init" in r["Kernel_Name"])
    if n_pass == 0:
        return [], {}
    for r in rows:
        if "pt_wf_" in r["Kernel_Name"]:
            k = short(r["Kernel_Name"])
            per_kernel[k] = per_kernel.get(k, 0.0) + value_of(r)
    total = sum(per_kernel.values())
    return [total / n_pass] * n_pass, {k: v / n_pass for k, v in per_kernel.items()}


def regime_summary(rows, regime, timed=20):
    """Span-based time per pass of a kernel trace: a pass is complete when its pt_wf_resolve (or pt_megakernel) ends; the steady
    figure is (end of the last pass - end of the last pass BEFORE the timed ones) / timed passes -- what bench.py's HIP events
    measure (the trace also holds bench.py's untimed priming and warm-up passes).  Also the sum of kernel durations per pass and the
    mean number of kernels running at once (sum / span), both over the timed passes only."""
    ends = sorted(int(r["End_Timestamp"]) for r in rows if "pt_wf_resolve" in r["Kernel_Name"] or "pt_megakernel" in r["Kernel_Name"])
    n = len(ends)
    res = {"regime": regime, "passes": n}
    if n > timed:
        t0, t1 = ends[-1 - timed], ends[-1]
        # kernels of the timed region: everything that ends after the last untimed pass has completed (bench.py synchronises there)
        busy = sum(int(r["End_Timestamp"]) - max(int(r["Start_Timestamp"]), t0) for r in rows if int(r["End_Timestamp"]) > t0)
        res.update({"timed_passes": timed, "span_ms": (t1 - t0) / 1e6, "steady_ms_per_pass": (t1 - t0) / 1e6 / timed,
                    "sum_of_kernel_ms_per_pass": busy / 1e6 / timed, "mean_kernels_in_flight": busy / max(1, t1 - t0),
                    "steady_note": f"(end of pass {n} - end of pass {n - timed}) / {timed} timed passes"})
    return {"trace": res}


def newest(pattern):
    """gpurun merges every call's files into gpurun_out/: several runs of one pass may lie side by side -- take the latest."""
    return sorted(glob.glob(pattern), key=os.path.getmtime, reverse=True)


def main():
    tag = sys.argv[1]
    workload = sys.argv[2] if len(sys.argv) > 2 else "sponza"
    base = os.path.join(ROOT, "gpurun_out")
    stats = newest(os.path.join(base, f"{tag}_trace", "*", "*_kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"))
    trace = newest(os.path.join(base, f"{tag}_trace", "*", "*_kernel_trace.csv"))
    out = {"tag": tag, "workload": workload}
    if trace:
        rows = [r for r in csv.DictReader(open(trace[0])) if "pt_" in r["Kernel_Name"]]
        per = {}
        for r in rows:
            per.setdefault(short(r["Kernel_Name"]), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
        out["kernel_ms"] = {k: {"calls": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v), "total": sum(v)} for k, v in per.items()}
        res = {}
        for r in rows:
            res[short(r["Kernel_Name"])] = {k: r.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size_X", "Grid_Size_X")}
        out["resources"] = res
        t0 = min(int(r["Start_Timestamp"]) for r in rows)
        t1 = max(int(r["End_Timestamp"]) for r in rows)
        n_pass = sum(1 for r in rows if "pt_wf_init" in r["Kernel_Name"] or "pt_megakernel" in r["Kernel_Name"] or "pt_wf_fused" in r["Kernel_Name"])
        out["passes"] = n_pass
        out["span_ms_per_pass"] = (t1 - t0) / 1e6 / max(1, n_pass)
        out.update(regime_summary(rows, "pipelined: the command bench.py's default run times (--steps 20 --warmup 5, the library's default number of passes in flight)"))
    serial = newest(os.path.join(base, f"{tag}_serial", "*", "*_kernel_trace.csv"))
    if serial:
        rows = [r for r in csv.DictReader(open(serial[0])) if "pt_" in r["Kernel_Name"]]
        ser = regime_summary(rows, "serialised: the same command with --passes-in-flight 1 (the kernels of a pass run back to back, nothing overlaps them): honest per-kernel durations")
        per = {}
        for r in rows:
            per.setdefault(short(r["Kernel_Name"]), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
        n = max(1, ser["trace"]["passes"])                                       # per-kernel means over ALL passes of the run (they run alone either way)
        ser["kernel_ms"] = {k: {"calls_per_pass": len(v) / n, "mean": sum(v) / len(v), "total_per_pass": sum(v) / n} for k, v in per.items()}
        out["serialised"] = ser
        st = newest(os.path.join(base, f"{tag}_serial", "*", "*_kernel_stats.csv"))
        if st:
            shutil.copy(st[0], os.path.join(ROOT, "profiles", f"{tag}_kernel_stats_serialised.csv"))
    pmc = {}
    for kind, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        f = newest(os.path.join(base, f"{tag}_{kind}", "*", "*_counter_collection.csv"))
        if not f:
            continue
        rows = [r for r in csv.DictReader(open(f[0])) if "pt_" in r["Kernel_Name"] and r["Counter_Name"] == counter]
        passes, per_kernel = per_pass(rows, lambda r: float(r["Counter_Value"]))
        pmc[counter] = {"KiB_per_pass": sum(passes) / max(1, len(passes)), "passes": len(passes), "KiB_per_pass_by_kernel": per_kernel}
    f = newest(os.path.join(base, f"{tag}_valu", "*", "*_counter_collection.csv"))
    if f:
        allrows = [r for r in csv.DictReader(open(f[0])) if "pt_" in r["Kernel_Name"]]
        for counter in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"):
            rows = [r for r in allrows if r["Counter_Name"] == counter]
            passes, per_kernel = per_pass(rows, lambda r: float(r["Counter_Value"]))
            if passes:
                pmc[counter] = {"per_pass": sum(passes) / len(passes), "passes": len(passes), "per_pass_by_kernel": per_kernel}
        if "SQ_INSTS_VALU" in pmc:
            # the second roofline of bench.py: wave-level VALU instructions of a pass against the MEASURED issue peak
            peak, peak_source = measured_valu_peak()
            json.dump({"workload": workload, "n_gpus": 1, "passes_profiled": pmc["SQ_INSTS_VALU"]["passes"],
                       "valu_wave_insts_per_launch": pmc["SQ_INSTS_VALU"]["per_pass"],
                       "by_kernel": pmc["SQ_INSTS_VALU"]["per_pass_by_kernel"],
                       "issue_peak_wave_insts_per_s": peak, "issue_peak_source": peak_source,
                       "source": "rocprofv3 --pmc SQ_INSTS_VALU (profiles/collect.sh %s), summed over the kernels of a pass; profiles/%s_pmc.json" % (tag, tag)},
                      open(os.path.join(ROOT, "profiles", "valu_insts.json"), "w"), indent=1)
    # where the wave-cycles go, per kernel: ACTIVE (issuing) / WAIT_INST_ANY (ready, waiting to issue) / WAIT_ANY (parked on s_waitcnt)
    f = newest(os.path.join(base, f"{tag}_wait", "*", "*_counter_collection.csv"))
    if f:
        agg = {}
        for r in csv.DictReader(open(f[0])):
            if "pt_" not in r["Kernel_Name"]:
                continue
            agg.setdefault(short(r["Kernel_Name"]), {}).setdefault(r["Counter_Name"], 0.0)
            agg[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
        waits = {}
        for k, v in agg.items():
            wc = v.get("SQ_WAVE_CYCLES", 0.0)
            if wc <= 0:
                continue
            waits[k] = {"wave_quad_cycles": wc, "waves": v.get("SQ_WAVES", 0.0), "valu_insts": v.get("SQ_INSTS_VALU", 0.0),
                        "frac_active_valu": v.get("SQ_ACTIVE_INST_VALU", 0.0) / wc, "frac_active_any": v.get("SQ_ACTIVE_INST_ANY", 0.0) / wc,
                        "frac_wait_issue": v.get("SQ_WAIT_INST_ANY", 0.0) / wc, "frac_wait_memory": v.get("SQ_WAIT_ANY", 0.0) / wc}
        pmc["wave_cycle_breakdown"] = {"by_kernel": waits, "note": "SQ_* in quad-cycles; ACTIVE_INST_ANY + WAIT_INST_ANY + WAIT_ANY ~ WAVE_CYCLES (profiled passes run serialised)"}
    f = newest(os.path.join(base, f"{tag}_lane", "*", "*_counter_collection.csv"))
    if f:
        agg = {}
        for r in csv.DictReader(open(f[0])):
            if "pt_" not in r["Kernel_Name"]:
                continue
            agg.setdefault(short(r["Kernel_Name"]), {}).setdefault(r["Counter_Name"], 0.0)
            agg[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
        pmc["lane_utilisation"] = {k: v["SQ_THREAD_CYCLES_VALU"] / (64.0 * v["SQ_ACTIVE_INST_VALU"]) for k, v in agg.items()
                                   if v.get("SQ_ACTIVE_INST_VALU", 0) > 0 and "SQ_THREAD_CYCLES_VALU" in v}
    # vector-memory path, L2 and dispatcher counters, summed per kernel over the profiled passes (one rocprofv3 run per block)
    mem = {}
    for name in ("ta", "td", "tcp", "tcc", "grbm", "spi"):
        f = newest(os.path.join(base, f"{tag}_{name}", "*", "*_counter_collection.csv"))
        if not f:
            continue
        for r in csv.DictReader(open(f[0])):
            if "pt_" not in r["Kernel_Name"]:
                continue
            k = short(r["Kernel_Name"])
            mem.setdefault(k, {}).setdefault(r["Counter_Name"], 0.0)
            mem[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in mem.items():
        cu_cycles = v.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 * 256.0          # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 256 CUs
        if cu_cycles > 0:
            if "TA_TA_BUSY_sum" in v:
                v["ta_busy_frac"] = v["TA_TA_BUSY_sum"] / cu_cycles
            if "TD_TD_BUSY_sum" in v:
                v["td_busy_frac"] = v["TD_TD_BUSY_sum"] / cu_cycles
        if v.get("TCP_TCC_READ_REQ_sum"):
            v["tcp_to_l2_read_latency_cycles"] = v.get("TCP_TCC_READ_REQ_LATENCY_sum", 0.0) / v["TCP_TCC_READ_REQ_sum"]
        if v.get("TCC_HIT_sum") is not None and (v.get("TCC_HIT_sum", 0) + v.get("TCC_MISS_sum", 0)) > 0:
            v["l2_hit_rate"] = v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"])
        if v.get("TA_FLAT_READ_WAVEFRONTS_sum") and v.get("TD_TD_BUSY_sum"):
            v["td_cycles_per_vector_load"] = v["TD_TD_BUSY_sum"] / v["TA_FLAT_READ_WAVEFRONTS_sum"]
    if mem:
        pmc["memory_path_and_dispatch"] = {"by_kernel": mem, "note": "sums over the profiled passes (passes run serialised under --pmc); *_frac against GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs"}
    out["pmc"] = pmc
    if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
        fetch = pmc["FETCH_SIZE"]["KiB_per_pass"] * 1024
        write = pmc["WRITE_SIZE"]["KiB_per_pass"] * 1024
        out["hbm_bytes_per_pass"] = {"fetch_raw": fetch, "fetch_x2_upper": 2 * fetch, "write": write,
                                     "total_raw": fetch + write, "total_upper": 2 * fetch + write}
        json.dump({"workload": workload, "n_gpus": 1, "bytes_per_launch": round(2 * fetch + write),
                   "source": "profiles/%s_pmc.json" % tag,
                   "note": "per pass (all kernels of the schedule): FETCH_SIZE x2 (gfx950 correction; upper estimate for scattered 16-B reads) "
                           "+ WRITE_SIZE, separate rocprofv3 --pmc passes (profiles/collect.sh %s)" % tag},
                  open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"), indent=1)
    json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_pmc.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

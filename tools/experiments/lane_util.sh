#!/bin/bash
# usage (GPU box): bash tools/experiments/lane_util.sh TAG -> profiles/TAG_lane_utilisation.json
# lanes active per VALU instruction = SQ_THREAD_CYCLES_VALU / (64 x SQ_INSTS_VALU-cycles), per kernel, one --pmc pass
TAG=$1
export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d gpurun_out/${TAG}_lanes -- python3 bench.py --steps 4 --warmup 1 --no-extra --no-cpu-baseline --no-alg-replay > gpurun_out/${TAG}_lanes.log 2>&1
python3 - "$TAG" <<'PY'
import csv, glob, json, sys, collections, os, re
tag = sys.argv[1]
f = sorted(glob.glob(f"gpurun_out/{tag}_lanes/*/*counter_collection.csv"), key=os.path.getmtime)[-1]
agg = collections.defaultdict(lambda: collections.Counter())
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "pt_" not in n: continue
    k = re.sub(r"\(.*", "", n.replace("void ", "").replace("(anonymous namespace)::", ""))
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
out = {}
for k, v in agg.items():
    if v["SQ_INSTS_VALU"] <= 0: continue
    # SQ_THREAD_CYCLES_VALU counts active lanes x cycles, SQ_ACTIVE_INST_VALU the cycles (both in the same 4-cycle units)
    out[k] = {"valu_insts": v["SQ_INSTS_VALU"], "salu_insts": v["SQ_INSTS_SALU"], "vmem_rd": v["SQ_INSTS_VMEM_RD"], "vmem_wr": v["SQ_INSTS_VMEM_WR"],
              "lanes_per_valu_inst": round(v["SQ_THREAD_CYCLES_VALU"] / max(1.0, v["SQ_ACTIVE_INST_VALU"]), 2),
              "lane_utilisation": round(v["SQ_THREAD_CYCLES_VALU"] / max(1.0, v["SQ_ACTIVE_INST_VALU"]) / 64.0, 4)}
json.dump({"tag": tag, "source": "rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_* over bench.py --steps 4 --warmup 1 (5 passes, kernels serialised)",
           "by_kernel": out}, open(f"profiles/{tag}_lane_utilisation.json", "w"), indent=1)
for k, v in out.items(): print(k, v)
PY

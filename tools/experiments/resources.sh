#!/bin/bash
# usage: [A_EXTRA=""] tools/experiments/resources.sh [-D flags]   (A_EXTRA: the extra flags of the default schedule's translation unit, csrc/Makefile; empty = the other one)
# usage: tools/experiments/resources.sh [-D flags]  -> per-kernel VGPRs / scratch / spills / occupancy of pt_wavefront.hip
cd "$(dirname "$0")/../../unity_webgpu_pathtracer_amd/csrc"
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fvisibility=hidden -I../../include -Wno-unused-function -Wno-unused-value -mllvm -amdgpu-sched-strategy=max-memory-clause -mllvm -amdgpu-use-amdgpu-trackers=1 ${A_EXTRA--mllvm -enable-post-misched=false}"
hipcc --offload-arch=gfx950 $FLAGS "$@" -c ${SRC:-pt_wavefront.hip} -o /tmp/res_$$.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re, sys
cur = None
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m: cur = m.group(1); print(); print(cur[:90], end=" ")
    for key in ("VGPRs:", "SGPRs:", "ScratchSize", "SGPRs Spill", "VGPRs Spill", "Occupancy", "LDS Size"):
        m = re.search(re.escape(key) + r"[^0-9]*(\d+)", line)
        if m and cur: print(key.split()[0].rstrip(":") + ("Sp" if "Spill" in key else "") + "=" + m.group(1), end=" ")
print()'
rm -f /tmp/res_$$.o

#!/bin/bash
# PT_CU_SPLIT experiment: shade launches on a stream confined to a CU partition (modes: pt_api.hip ensure_wavefront)
run() { python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-alg-replay --latency-steps 0 $EXTRA 2>gpurun_out/cusplit.err | python -c 'import sys, json
for l in sys.stdin:
    if l.startswith("{"): j = json.loads(l); print("Mrays/s", j["value"], "ms", j["ms_per_step"], "crc", j["config"].get("frame_crc32"))' || tail -3 gpurun_out/cusplit.err; }
export GPU_MAX_HW_QUEUES=32
echo "== off (32 queues)"; run
for m in "$@"; do echo "== PT_CU_SPLIT=$m"; PT_CU_SPLIT=$m run; done
echo "== off (again)"; run

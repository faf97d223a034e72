#!/bin/bash
# usage: tools/experiments/main_ab.sh [variant names...]  -> bench.py main workload (20 steps) through the default build (first and last) and the named variants
run() { python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-alg-replay 2>/dev/null | python -c 'import sys, json
for l in sys.stdin:
    if l.startswith("{"): j = json.loads(l); print("Mrays/s", j["value"], "ms", j["ms_per_step"], "crc", j.get("frame_crc32"))'; }
echo "== default"; run
for n in "$@"; do echo "== $n"; PT_PLUGIN=$PWD/tools/experiments/variants/$n.so run; done
echo "== default (again)"; run

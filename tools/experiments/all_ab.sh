#!/bin/bash
# usage: tools/experiments/all_ab.sh [variant names...]  -> bench.py with the secondary workloads (Cornell: megakernel; bunny-class) through the default build and the named variants
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alg-replay --latency-steps 0 2>/dev/null | python -c 'import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        j = json.loads(l); print("sponza", j["value"], "|", " | ".join(k + " " + str(v.get("value", v.get("mrays_per_s"))) for k, v in j.get("other_workloads", {}).items()))'; }
echo "== default"; run
for n in "$@"; do echo "== $n"; PT_PLUGIN=$PWD/tools/experiments/variants/$n.so run; done
echo "== default (again)"; run

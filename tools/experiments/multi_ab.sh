#!/bin/bash
# usage: tools/experiments/multi_ab.sh [variants...]: main + bunny + Cornell (bench.py), the HAS_TLAS scene and schedule 4 (one pass in flight) per build
one() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alg-replay --latency-steps 0 2>/dev/null | python -c 'import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        j = json.loads(l); print("sponza", j["value"], "|", " | ".join(k + " " + str(v.get("mrays_per_s")) for k, v in j.get("other_workloads", {}).items()), end=" | ")'
  python tools/experiments/tlas_perf.py tlas 2>&1 | grep "^tlas" | sed 's/.*pass, /tlas /' | tr '\n' ' '
  python bench.py --schedule 4 --passes-in-flight 1 --steps 12 --warmup 3 --no-extra --no-cpu-baseline --no-alg-replay --latency-steps 0 2>/dev/null | python -c 'import sys, json
for l in sys.stdin:
    if l.startswith("{"): j = json.loads(l); print("| schedule 4:", j["ms_per_step"], "ms")'; }
echo "== default"; one
for n in "$@"; do echo "== $n"; PT_PLUGIN=$PWD/tools/experiments/variants/$n.so one; done

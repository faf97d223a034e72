#!/bin/bash
# schedule 4 (fused persistent wavefront), one and two passes in flight, default build vs a variant: usage sched4_ab.sh [variant]
run() { for cfg in "4 1" "4 2" "1 0"; do set -- $cfg
  python bench.py --schedule $1 --passes-in-flight $2 --no-extra --no-cpu-baseline --no-alg-replay 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('schedule $1 passes_in_flight', d['config']['passes_in_flight'], ': %.0f Mrays/s  %.2f ms/pass  latency %.2f ms  crc %d' % (d['value'], d['ms_per_step'], d['latency_mode']['ms_per_pass'], d['config']['frame_crc32']))"
done; }
echo "== default"; run
for n in "$@"; do echo "== $n"; PT_PLUGIN=$PWD/tools/experiments/variants/$n.so run; done

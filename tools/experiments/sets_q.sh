#!/bin/bash
one() { python bench.py --no-extra --no-cpu-baseline --no-alg-replay --latency-steps 0 $1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f Mrays/s %.2f ms (in flight %d)' % (d['value'], d['ms_per_step'], d['config']['passes_in_flight']))"; }
echo -n "default 12 sets q16: "; one ""
for cfg in "s16 24 16" "s16 32 16" "s20 32 20" "s20 24 20"; do set -- $cfg
  echo -n "$1 queues $2 sets $3: "; GPU_MAX_HW_QUEUES=$2 PT_PLUGIN=$PWD/tools/experiments/variants/$1.so one "--passes-in-flight $3"
done
echo -n "default 12 sets q16: "; one ""
echo -n "default, 8 sets: "; one "--passes-in-flight 8"
echo -n "default, 10 sets: "; one "--passes-in-flight 10"

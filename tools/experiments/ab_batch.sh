#!/bin/bash
one() { python bench.py --no-extra --no-cpu-baseline --no-alg-replay --latency-steps 0 $1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f Mrays/s %.2f ms' % (d['value'], d['ms_per_step']))"; }
for i in 1 2; do
  for b in 1 2 4; do echo -n "batch $b: "; one "--batch $b"; done
  echo -n "batch 2, 6 in flight: "; one "--batch 2 --passes-in-flight 6"
  echo -n "batch 4, 4 in flight: "; one "--batch 4 --passes-in-flight 4"
done

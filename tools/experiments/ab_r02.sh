#!/bin/bash
# A/B on ONE box: the round-2 tree (tools/experiments/variants/r02tree, git 3014073) against the working tree, alternating
one() { python bench.py --no-extra --no-cpu-baseline --no-alg-replay $1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f Mrays/s %.2f ms' % (d['value'], d['ms_per_step']))"; }
for i in 1 2 3; do
  echo -n "r02: "; (cd tools/experiments/variants/r02tree && one "")
  echo -n "now: "; one "--latency-steps 0"
done

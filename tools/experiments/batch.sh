#!/bin/bash
for b in 1 2 4 8; do
  python bench.py --schedule 1 --batch $b --steps 24 --warmup 8 --no-extra --no-cpu-baseline --no-alg-replay --latency-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('batch $b: %.0f Mrays/s  %.2f ms/pass  crc %d' % (d['value'], d['ms_per_step'], d['config']['frame_crc32']))"
done

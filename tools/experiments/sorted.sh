#!/bin/bash
for v in 0 2 1 0 2; do
  PT_WF_SHADE_SORTED=$v python bench.py --schedule 1 --no-extra --no-cpu-baseline --no-alg-replay 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('shade sorted $v: %.0f Mrays/s  %.2f ms/pass  latency %.2f ms crc %d' % (d['value'], d['ms_per_step'], d['latency_mode']['ms_per_pass'], d['config']['frame_crc32']))"
done

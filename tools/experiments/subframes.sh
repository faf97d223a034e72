#!/bin/bash
# latency of one pass (host sync after every pass) against the number of sub-frames (PTSetSubFrames); schedule 1
for s in 1 2 3 4 6 8 12; do
  python bench.py --schedule 1 --latency-sub-frames $s --steps 4 --warmup 14 --latency-steps 8 --no-extra --no-cpu-baseline --no-alg-replay 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
l=d['latency_mode']
print('sub-frames $s: latency %.2f ms/pass (%s)  crc %d' % (l['ms_per_pass'], l['ms_per_pass_all'], d['config']['frame_crc32']))"
done

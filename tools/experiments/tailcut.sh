#!/bin/bash
# tail cut thresholds: throughput (12 passes in flight) and latency (host sync per pass), schedule 1
for t in 0 256 2048 8192 32768 131072; do
  PT_WF_TAIL_CUT=$t python bench.py --schedule 1 --no-extra --no-cpu-baseline --no-alg-replay 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('tail cut $t: %.0f Mrays/s  %.2f ms/pass  latency %.2f ms %s crc %d' % (d['value'], d['ms_per_step'], d['latency_mode']['ms_per_pass'], d['latency_mode']['ms_per_pass_all'][2:], d['config']['frame_crc32']))"
done

#!/bin/bash
# What does a rocprofv3 --pmc run leave behind that slows the NEXT run in the same job?  Clocks / performance level before and after.
export TMPDIR=/tmp
B="python3 bench.py --workload bunny --steps 10 --warmup 3 --latency-steps 0 --no-extra --no-cpu-baseline --no-alg-replay"
show() { rocm-smi --showperflevel --showclocks 2>/dev/null | grep -E "Performance Level|sclk|mclk|fclk" | head -6 | tr '\n' ';'; echo; }
val() { python3 -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('   bunny %.0f Mrays/s %.2f ms/pass' % (d['value'], d['ms_per_step']))"; }
echo "before:"; show; $B 2>/dev/null | val
echo "kernel-trace only:"; rocprofv3 --kernel-trace --output-format csv -d gpurun_out/leak_kt -- $B > /dev/null 2>&1; show; $B 2>/dev/null | val
echo "after --pmc SQ_INSTS_VALU:"; rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace --output-format csv -d gpurun_out/leak_pmc -- $B > /dev/null 2>&1; show; $B 2>/dev/null | val
sleep 20; echo "20 s later:"; show; $B 2>/dev/null | val

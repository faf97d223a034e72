#!/bin/bash
# usage (GPU box): bash tools/experiments/after_tests.sh -- which GPU test file leaves the device slower for the next process?
b() { python3 bench.py --no-extra --no-cpu-baseline --no-alg-replay | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('   bench', j['value'])"; }
b; b
for f in tests/test_*.py; do
  n=$(python -m pytest $f -m gpu -q 2>&1 | tail -1)
  echo "$f: $n"
  case "$n" in *passed*) b;; esac
done

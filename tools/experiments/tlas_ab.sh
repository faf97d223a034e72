#!/bin/bash
# usage: [PT_SETS=n] tools/experiments/tlas_ab.sh [variant names...]  -> HAS_TLAS scene through the default build (first and last) and the named variants
run() { python tools/experiments/tlas_perf.py tlas 2>&1 | grep "^tlas"; }
echo "== default"; run
for n in "$@"; do echo "== $n"; PT_PLUGIN=$PWD/tools/experiments/variants/$n.so run; done
echo "== default (again)"; run

#!/bin/bash
echo "== default"; python tools/experiments/tlas_counts.py 2>&1 | grep "^tlas"
for so in tools/experiments/variants/t_*.so; do echo "== $so"; PT_PLUGIN=$PWD/$so python tools/experiments/tlas_counts.py 2>&1 | grep "^tlas"; done

for n in 3 6 8; do echo "== sets $n"; PT_PLUGIN=$PWD/scratch/variants/sets$n.so timeout -k 10 300 python scratch/shard_emulate2.py 2>&1 | grep "sched 1"; done

PT_PLUGIN=$PWD/scratch/variants/g1.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not schedule0 and not full_size" 2>&1 | tail -3
timeout -k 10 600 python scratch/variants.py

export TMPDIR=/tmp
for v in d0_s8 d8_s8; do
export PT_PLUGIN=$PWD/tools/experiments/variants/$v.so
CMD="python3 bench.py --steps 1 --warmup 1 --no-extra --no-cpu-baseline --no-alg-replay"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/def_$v -- $CMD > gpurun_out/def_$v.log 2>&1
done
python3 tools/experiments/pmc_sum.py gpurun_out/def_d0_s8 gpurun_out/def_d8_s8 2>&1 | grep -v "rocclr\|at::native\|fold\|resolve\|init\|cleanup\|shade"

export TMPDIR=/tmp
for v in g0 g16; do
export PT_PLUGIN=$PWD/tools/experiments/variants/$v.so
CMD="python3 bench.py --steps 1 --warmup 1 --no-extra --no-cpu-baseline --no-alg-replay"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d gpurun_out/gath_$v -- $CMD > gpurun_out/gath_$v.log 2>&1
done
python3 tools/experiments/pmc_sum.py gpurun_out/gath_g0 gpurun_out/gath_g16 2>&1 | grep -v "rocclr\|at::native\|fold\|resolve\|init\|cleanup"

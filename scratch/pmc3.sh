export TMPDIR=/tmp
CMD="python3 bench.py --steps 1 --warmup 1 --no-extra --no-cpu-baseline --no-alg-replay"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_SALU --kernel-trace --output-format csv -d gpurun_out/pmc_sq5 -- $CMD > gpurun_out/pmc_sq5.log 2>&1
echo done

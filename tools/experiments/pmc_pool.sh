export TMPDIR=/tmp
for s in 1 4; do
CMD="python3 bench.py --schedule $s --steps 1 --warmup 1 --no-extra --no-cpu-baseline --no-alg-replay"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d gpurun_out/pool_a$s -- $CMD > gpurun_out/pool_a$s.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/pool_b$s -- $CMD > gpurun_out/pool_b$s.log 2>&1
done
python3 tools/experiments/pmc_sum.py gpurun_out/pool_a1 gpurun_out/pool_a4 gpurun_out/pool_b1 gpurun_out/pool_b4 > gpurun_out/pool_pmc.txt 2>&1
echo done

export TMPDIR=/tmp
CMD="python3 bench.py --steps 1 --warmup 1 --no-extra --no-cpu-baseline --no-alg-replay"
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_sq3 -- $CMD > gpurun_out/pmc_sq3.log 2>&1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum --kernel-trace --output-format csv -d gpurun_out/pmc_tcp -- $CMD > gpurun_out/pmc_tcp.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_CVT --kernel-trace --output-format csv -d gpurun_out/pmc_sq4 -- $CMD > gpurun_out/pmc_sq4.log 2>&1
echo done

export TMPDIR=/tmp
export PT_PLUGIN=$PWD/tools/experiments/variants/base.so
CMD="python3 bench.py --steps 4 --warmup 1 --no-extra --no-cpu-baseline --no-alg-replay"
timeout -k 10 150 rocprofv3 --pmc SPI_RA_TMP_STALL_CSN SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_VGPR_SIMD_FULL_CSN SPI_RA_LDS_CU_FULL_CSN --kernel-trace --output-format csv -d gpurun_out/spi1 -- $CMD > gpurun_out/spi1.log 2>&1; python3 tools/experiments/pmc_sum.py gpurun_out/spi1
timeout -k 10 150 rocprofv3 --pmc SPI_RA_RES_STALL_CSN SPI_RA_REQ_NO_ALLOC_CSN SPI_CSN_BUSY SPI_CSN_WAVE --kernel-trace --output-format csv -d gpurun_out/spi2 -- $CMD > gpurun_out/spi2.log 2>&1; python3 tools/experiments/pmc_sum.py gpurun_out/spi2

#!/bin/bash
out=gpurun_out/r03e; mkdir -p $out
tools/micro/valu_rates.bin > $out/valu_rates.txt 2>&1; cat $out/valu_rates.txt
tools/ab_kernel.sh 2 abl6 tag6 abl5p tag5p -- --no-renderer-only > $out/ab.txt 2>&1; cat $out/ab.txt
V=$PWD/gaussian-splatting-lightning_amd/variants
GSPL_HIP_LIB=$V/libgspl_hip_tag6.so timeout 600 python -m pytest tests/test_hip_parity.py tests/test_metric_point_parity.py tests/test_backward_spread.py -q -m gpu -p no:cacheprovider -x > $out/pytest_tag6.log 2>&1; tail -5 $out/pytest_tag6.log
GSPL_HIP_LIB=$V/libgspl_hip_tag6.so tools/pmc_quick.sh "" composite_bwd SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY > $out/pmc_tag6_a.txt 2>&1; cat $out/pmc_tag6_a.txt

#!/bin/bash
out=gpurun_out/r03g; mkdir -p $out
V=$PWD/gaussian-splatting-lightning_amd/variants
GSPL_HIP_LIB=$V/libgspl_hip_pk5.so timeout 600 python -m pytest tests/test_hip_parity.py tests/test_metric_point_parity.py tests/test_backward_spread.py tests/test_renderers_gpu.py -q -m gpu -p no:cacheprovider > $out/pytest_pk5.log 2>&1; tail -3 $out/pytest_pk5.log
tools/ab_kernel.sh 2 pk5 pk6c48 pk4 -- --no-renderer-only > $out/ab.txt 2>&1; cat $out/ab.txt
GSPL_HIP_LIB=$V/libgspl_hip_pk5.so tools/pmc_quick.sh "" composite_bwd SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY > $out/pmc_pk5_a.txt 2>&1; cat $out/pmc_pk5_a.txt
GSPL_HIP_LIB=$V/libgspl_hip_pk5.so tools/pmc_quick.sh "" composite_bwd SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_ADDR_CONFLICT > $out/pmc_pk5_b.txt 2>&1; cat $out/pmc_pk5_b.txt

#!/bin/bash
out=gpurun_out/r03i; mkdir -p $out
V=$PWD/gaussian-splatting-lightning_amd/variants
tools/ab_kernel.sh 2 lpt4 lpt4p64 lpt5p64 -- --no-renderer-only > $out/ab.txt 2>&1; cat $out/ab.txt
GSPL_HIP_LIB=$V/libgspl_hip_lpt4p64.so timeout 600 python -m pytest tests/test_hip_parity.py tests/test_metric_point_parity.py tests/test_backward_spread.py -q -m gpu -p no:cacheprovider -x > $out/pytest.log 2>&1; tail -3 $out/pytest.log
GSPL_HIP_LIB=$V/libgspl_hip_lpt4p64.so tools/pmc_quick.sh "" composite_bwd SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY > $out/pmc_a.txt 2>&1; cat $out/pmc_a.txt

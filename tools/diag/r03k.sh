#!/bin/bash
out=gpurun_out/r03k; mkdir -p $out
tools/ab_kernel.sh 3 old -- --no-renderer-only > $out/ab.txt 2>&1; cat $out/ab.txt
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_metric_point_parity.py tests/test_backward_spread.py tests/test_renderers_gpu.py -q -m gpu -p no:cacheprovider > $out/pytest.log 2>&1; tail -3 $out/pytest.log

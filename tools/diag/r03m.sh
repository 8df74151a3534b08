#!/bin/bash
out=gpurun_out/r03m; mkdir -p $out
timeout 1200 python -m pytest tests/test_locked_parity.py -q -m gpu -p no:cacheprovider -s > $out/pytest.log 2>&1; grep -E "locked|tail|passed|failed|Error|assert" $out/pytest.log | head -60

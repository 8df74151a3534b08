#!/bin/bash
# The ONE scratch script of a gpurun call (rewritten per call; outputs under gpurun_out/<tag>/, the keepers are copied to profiles/).
tag=${1:-r07g}
cd /root/repo
O=gpurun_out/$tag; mkdir -p $O
timeout 600 python -m pytest tests/test_fused_backward_adam.py -q -m gpu > $O/${tag}_fused_adam_tests.txt 2>&1; grep -E "passed|failed|part at step|AssertionError" $O/${tag}_fused_adam_tests.txt | cut -c1-600
timeout 1500 python -m pytest tests/test_locked_parity.py -q -m gpu -s > $O/${tag}_locked.txt 2>&1; grep -E "^\[locked.*(flagged|conditioned)|passed|failed|AssertionError:" $O/${tag}_locked.txt | cut -c1-330

#!/bin/bash
# default exchange "auto" + three-node step: every test that touches the sharded renderer, the bench contract, one-rank RCCL
out=gpurun_out/r04c
mkdir -p $out
timeout 800 python -m pytest tests/test_records.py tests/test_renderers_gpu.py tests/test_distributed_renderer.py tests/test_rccl_single_rank.py \
    tests/test_bench_contract.py tests/test_training_loop.py -x -q -m gpu 2>&1 | tail -8 > $out/tests.txt
cat $out/tests.txt

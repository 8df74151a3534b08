#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tools/micro/adam_bandwidth.py 2>&1 | tail -8 | tee gpurun_out/r06j_adam_bandwidth.txt

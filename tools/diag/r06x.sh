#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tools/micro/host_step.py 300 0 2>&1 | tail -12 | tee gpurun_out/r06x_host_step.txt

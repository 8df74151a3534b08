#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
V=$GRAFT_REPO_ROOT/gaussian-splatting-lightning_amd/variants
timeout 600 python -m pytest tests/test_loss.py -x -q -m gpu 2>&1 | tail -2
GSPL_HIP_LIB=$V/libgspl_hip_loss_both3.so timeout 600 python -m pytest tests/test_loss.py -x -q -m gpu 2>&1 | tail -2
for rep in 1 2; do
for v in base fwd both both3; do
  if [ $v = both ]; then unset GSPL_HIP_LIB; else export GSPL_HIP_LIB=$V/libgspl_hip_loss_$v.so; fi
  echo -n "$v: "; python tools/micro/loss_time.py 300 2>&1 | tail -1
done; done | tee gpurun_out/r06t_loss_ahead.txt

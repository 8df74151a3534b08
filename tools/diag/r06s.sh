#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
V=$GRAFT_REPO_ROOT/gaussian-splatting-lightning_amd/variants
for rep in 1 2; do
for v in base fwd both both3; do
  if [ $v = both ]; then unset GSPL_HIP_LIB; else export GSPL_HIP_LIB=$V/libgspl_hip_loss_$v.so; fi
  rm -rf /tmp/prof
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $GRAFT_REPO_ROOT/tools/micro/loss_time.py 200 > /tmp/log.txt 2>&1)
  f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
  echo "== $v"; python tools/prof_summary.py stats $f 1 | grep "loss_"
done; done | tee gpurun_out/r06s_loss_ahead_kernels.txt

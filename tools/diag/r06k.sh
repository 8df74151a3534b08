#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
V=$GRAFT_REPO_ROOT/gaussian-splatting-lightning_amd/variants
for v in base u4 adj2 adj4 ntp g4096 g2048adj4 g65536 base; do
  if [ $v = base ]; then unset GSPL_HIP_LIB; else export GSPL_HIP_LIB=$V/libgspl_hip_adam_$v.so; fi
  echo "== $v"; python tools/micro/adam_bandwidth.py 2>&1 | grep "N=" | cut -c1-60
done | tee gpurun_out/r06k_adam_variants.txt

#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
V=$GRAFT_REPO_ROOT/gaussian-splatting-lightning_amd/variants
for v in tile4096 tile8192; do
  GSPL_HIP_LIB=$V/libgspl_hip_$v.so timeout 600 python -m pytest tests/test_sort.py tests/test_hip_parity.py -x -q -m gpu -k "sort or binning or isect or bin" 2>&1 | tail -2
done
for w in S-1080p-6M S-1080p-1M; do
for v in base tile4096 tile8192; do
  if [ $v = base ]; then unset GSPL_HIP_LIB; else export GSPL_HIP_LIB=$V/libgspl_hip_$v.so; fi
  rm -rf /tmp/prof
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $GRAFT_REPO_ROOT/bench.py --workload $w --steps 12 --warmup 3 --no-cpu-baseline --no-renderer-only --loop none --no-stage-rooflines --no-workload-stats > /tmp/log.txt 2>&1)
  f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
  echo "== $w $v"; python tools/prof_summary.py stats $f 1 | grep "unsigned int\|TOTAL"
done; done | tee gpurun_out/r06o_u32_tile_sizes.txt

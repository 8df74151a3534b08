#!/bin/bash
# round 4, session 2, call b: record prefetch one round ahead in the compositing kernels (variants), parity + kernel stats
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
V=$GRAFT_REPO_ROOT/gaussian-splatting-lightning_amd/variants
for v in bothpf; do
  GSPL_HIP_LIB=$V/libgspl_hip_$v.so timeout 900 python -m pytest tests/test_locked_parity.py tests/test_hip_parity.py tests/test_backward_spread.py -x -q -m gpu 2>&1 | tail -4 > $O/r06b_tests_$v.txt
  tail -2 $O/r06b_tests_$v.txt
done
for v in base fwdpf bwdpf bwdpf4 base2; do
  rm -rf /tmp/prof
  if [ $v = base ] || [ $v = base2 ]; then unset GSPL_HIP_LIB; else export GSPL_HIP_LIB=$V/libgspl_hip_$v.so; fi
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-renderer-only --loop none --no-stage-rooflines --no-workload-stats > /tmp/log_$v.txt 2>&1)
  f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); python tools/prof_summary.py stats $f 35 $O/r06b_kstats_$v.csv > /dev/null
  echo "== $v"; grep "composite_\|TOTAL" $O/r06b_kstats_$v.csv; tail -1 /tmp/log_$v.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step', d['ms_per_step'])"
done

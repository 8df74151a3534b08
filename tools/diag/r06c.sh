#!/bin/bash
# round 4, session 2, call c: early depth keys (span pass beside the depth sort) — full GPU suite, A/B kernel sequences
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 ) > $O/r06c_tests.txt 2>&1
tail -8 $O/r06c_tests.txt
for v in off on off2 on2; do
  rm -rf /tmp/prof
  case $v in off*) export GSPL_EARLY_KEYS=0;; *) unset GSPL_EARLY_KEYS;; esac
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-renderer-only --loop none --no-stage-rooflines --no-workload-stats > /tmp/log_$v.txt 2>&1)
  f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); python tools/prof_summary.py stats $f 35 $O/r06c_kstats_$v.csv > /dev/null
  f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1); python tools/prof_summary.py seq $f composite_fwd $O/r06c_seq_$v.txt > /dev/null
  echo "== $v"; tail -1 /tmp/log_$v.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step', d['ms_per_step'], 'img/s', d['value'])"; tail -1 $O/r06c_seq_$v.txt
done
unset GSPL_EARLY_KEYS
for i in 1 2; do
GSPL_EARLY_KEYS=0 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --loop none --no-stage-rooflines --no-workload-stats --no-renderer-only 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('off ms/step', d['ms_per_step'])"
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --loop none --no-stage-rooflines --no-workload-stats --no-renderer-only 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('on  ms/step', d['ms_per_step'])"
done

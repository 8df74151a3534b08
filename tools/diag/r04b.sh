#!/bin/bash
# padded format in the three-node sharded step: parity tests, then counted vs padded step time (W = 1, and one-rank RCCL group)
out=gpurun_out/r04b
mkdir -p $out
timeout 420 python -m pytest tests/test_records.py tests/test_renderers_gpu.py tests/test_distributed_renderer.py tests/test_rccl_single_rank.py \
    -x -q -m gpu -k "two_phase or pad_kernel or distributed or sharded or rccl" 2>&1 | tail -15 > $out/tests.txt
cat $out/tests.txt
F="--parallelism sharded --no-cpu-baseline --no-stage-rooflines --no-workload-stats"
show='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["value"], d["step_ms"]["p50"], d["step_ms"]["p99"], d["config"]["parallelism"][-90:])'
for i in 1 2; do
  python bench.py $F 2>/dev/null | tee -a $out/bench_counted.jsonl | python -c "$show" counted
  python bench.py $F --exchange padded 2>/dev/null | tee -a $out/bench_padded.jsonl | python -c "$show" padded
  python bench.py $F --init-dist 2>/dev/null | tee -a $out/bench_counted_rccl1.jsonl | python -c "$show" counted-rccl1
  python bench.py $F --init-dist --exchange padded 2>/dev/null | tee -a $out/bench_padded_rccl1.jsonl | python -c "$show" padded-rccl1
done
python bench.py $F --init-dist --exchange auto 2>/dev/null | tee -a $out/bench_auto_rccl1.jsonl | python -c "$show" auto-rccl1

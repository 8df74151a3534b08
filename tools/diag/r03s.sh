#!/bin/bash
# round 3: split SH parameters + deferred shs_rest update: tests, then the bench with and without the overlap
mkdir -p gpurun_out
python -m pytest tests/test_adam.py -x -q -m gpu -s 2>&1 | grep -v "^$" | tail -25 > gpurun_out/r03s_tests.txt
tail -5 gpurun_out/r03s_tests.txt
for blocks in 256 512 1024 2048; do
GSPL_ADAM_DEFERRED_BLOCKS=$blocks python bench.py --steps 200 --warmup 20 --no-stage-rooflines --no-workload-stats --no-cpu-baseline --no-renderer-only > gpurun_out/r03s_bench_overlap_$blocks.json 2> gpurun_out/r03s_bench_overlap.err
done
python bench.py --steps 200 --warmup 20 --no-stage-rooflines --no-workload-stats --no-overlap-sh-update --no-cpu-baseline --no-renderer-only > gpurun_out/r03s_bench_plain.json 2> gpurun_out/r03s_bench_plain.err
python - <<'PY'
import json
for n in ("overlap_256", "overlap_512", "overlap_1024", "overlap_2048", "plain"):
    try:
        d = json.loads(open(f"gpurun_out/r03s_bench_{n}.json").read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], d["step_ms"])
    except Exception as e:
        print(n, "failed", e)
PY

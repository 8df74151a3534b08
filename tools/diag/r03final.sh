#!/bin/bash
# whole GPU suite + smoke + bench line at HEAD (one fresh lease); tails kept under gpurun_out/<tag>/
tag=${1:-final1}
mkdir -p gpurun_out/$tag
git_head=$(cat .git_head 2>/dev/null)
python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/$tag/gputest_tail.txt
cat gpurun_out/$tag/gputest_tail.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 > gpurun_out/$tag/smoke_tail.txt
cat gpurun_out/$tag/smoke_tail.txt
python bench.py 2>/dev/null | tail -1 > gpurun_out/$tag/bench.json
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/$tag/bench_driver_form.json
python - $tag <<'PY'
import json, sys
for f in ("bench", "bench_driver_form"):
    d = json.loads(open(f"gpurun_out/{sys.argv[1]}/{f}.json").read())
    print(f, d["value"], d["ms_per_step"], d["step_ms"], d["roofline"]["frac"], d["roofline"]["traffic"], d["speculation"])
PY

#!/bin/bash
# whole GPU suite + smoke + bench line at HEAD (one lease); tails kept
mkdir -p gpurun_out/r03v
python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r03v/gputest_tail.txt
cat gpurun_out/r03v/gputest_tail.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6 > gpurun_out/r03v/smoke_tail.txt
cat gpurun_out/r03v/smoke_tail.txt
python bench.py 2>/dev/null | tail -1 > gpurun_out/r03v/bench.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03v/bench.json").read())
print(d["value"], d["ms_per_step"], d["step_ms"], d["roofline"]["frac"], d["roofline"]["traffic"], d["speculation"])
print(json.dumps(d["stage_rooflines"])[:1500])
PY

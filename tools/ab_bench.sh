#!/bin/bash
# usage (on the GPU box): tools/ab_bench.sh <variant .so> [rounds] [bench args]  -> interleaved A/B of the in-tree library against a variant
# (medians of ms/step over short runs; long runs drift with the clocks).
var=$1; rounds=${2:-6}; shift; shift
run() { local lib=$1; if [ -z "$lib" ]; then lib=$(pwd)/gaussian-splatting-lightning_amd/libgspl_hip.so; fi; GSPL_HIP_LIB=$lib python bench.py --steps 30 --warmup 10 --no-cpu-baseline "${@:2}" 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
a=(); b=()
for i in $(seq $rounds); do a+=($(run "" "$@")); b+=($(run $var "$@")); done
python - "${a[*]}" "${b[*]}" <<'PY'
import sys, statistics as st
a = [float(x) for x in sys.argv[1].split()]; b = [float(x) for x in sys.argv[2].split()]
print("in-tree ms/step median %.4f  (min %.4f)   variant median %.4f  (min %.4f)   variant/in-tree %.4f" % (st.median(a), min(a), st.median(b), min(b), st.median(b) / st.median(a)))
PY

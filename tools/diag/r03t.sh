#!/bin/bash
# kernel timeline of the step with the deferred shs_rest update
mkdir -p gpurun_out/r03t
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_t -- python /root/repo/bench.py --steps 40 --warmup 16 --no-cpu-baseline --no-renderer-only --no-stage-rooflines --no-workload-stats > /tmp/log_t.txt 2>&1
tail -2 /tmp/log_t.txt
f=$(find /tmp/prof_t -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(list(rows[0].keys()))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last 3 steps: find the last 3 composite_fwd launches
idx = [i for i, r in enumerate(rows) if "composite_fwd_kernel" in r["Kernel_Name"]]
lo = idx[-4]
t0 = int(rows[lo]["Start_Timestamp"])
out = open("/root/repo/gpurun_out/r03t/timeline.txt", "w")
for r in rows[lo:]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].split("(")[0][:70]
    out.write(f"{s/1e3:10.1f} {e/1e3:10.1f} {(e-s)/1e3:8.1f}  q={r.get('Queue_Id','?'):>3} s={r.get('Stream_Id','?'):>3}  {name}\n")
out.close()
PY

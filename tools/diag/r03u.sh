#!/bin/bash
# kernel timeline of the sharded renderer's step at W = 1 (gaps = host / read-backs)
mkdir -p gpurun_out/r03u
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof_u -- python /root/repo/bench.py --parallelism sharded --steps 40 --warmup 16 --no-cpu-baseline --no-renderer-only --no-stage-rooflines --no-workload-stats > /tmp/log_u.txt 2>&1
tail -1 /tmp/log_u.txt | cut -c1-400
f=$(find /tmp/prof_u -name "*kernel_trace.csv" | head -1)
g=$(find /tmp/prof_u -name "*memory_copy_trace.csv" | head -1)
python - "$f" "$g" <<'PY'
import csv, sys
rows = [dict(r, kind="K") for r in csv.DictReader(open(sys.argv[1]))]
try:
    for r in csv.DictReader(open(sys.argv[2])):
        rows.append({"Kernel_Name": "MEMCPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", "")), "Start_Timestamp": r["Start_Timestamp"], "End_Timestamp": r["End_Timestamp"], "Queue_Id": "-", "Stream_Id": r.get("Stream_Id", "-")})
except Exception as e:
    print("no memcpy trace", e)
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "composite_fwd_kernel" in r["Kernel_Name"]]
lo = idx[-3]
hi = idx[-1]
t0 = int(rows[lo]["Start_Timestamp"])
out = open("/root/repo/gpurun_out/r03u/timeline_sharded.txt", "w")
prev_end = None
for r in rows[lo:hi + 1]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    prev_end = max(prev_end or 0, e)
    name = r["Kernel_Name"].split("(")[0][:80]
    out.write(f"{s/1e3:10.1f} {(e-s)/1e3:8.1f} gap {gap:7.1f}  q={r.get('Queue_Id','?'):>3} s={r.get('Stream_Id','?'):>3}  {name}\n")
out.close()
PY

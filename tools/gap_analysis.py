#!/usr/bin/env python
"""Who is late at the gaps of a step: the host or the device?

  python tools/gap_analysis.py <hip_api_trace.csv> <kernel_trace.csv> <anchor kernel substring> [out.txt]

rocprofv3 --hip-trace --kernel-trace of a bench run (no counters): for the LAST step (between the last two launches of the anchor
kernel) every kernel with its idle gap on the device and, matched by correlation id, the host call that launched it — the thread it came
from and `lead_us` = the kernel's start on the device minus the end of its launch call on the host.  A kernel launched long before it
could start (lead >> 0) was waiting in the queue: a gap in front of it is the device's; lead ~ 0 means the device was waiting for the host.
Also listed: the host's other HIP calls (events, memsets, queries) that fall inside each gap."""
import csv
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from prof_summary import short  # noqa: E402


def main(api_path, ker_path, anchor, out=None):
    api = list(csv.DictReader(open(api_path)))
    by_corr = {}
    for r in api:
        by_corr.setdefault(r["Correlation_Id"], []).append(r)
    ker = sorted(csv.DictReader(open(ker_path)), key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(ker) if anchor in r["Kernel_Name"]]
    a, b = idx[-2], idx[-1]
    t0 = int(ker[a]["Start_Timestamp"])
    api_sorted = sorted(api, key=lambda r: int(r["Start_Timestamp"]))
    lines = ["start_us   dur_us  gap_us  lead_us   thread  kernel   | other HIP calls the host made between the previous kernel's launch call and this one's"]
    prev_end, prev_call_end = None, None
    quiet = ("hipGetLastError", "hipPeekAtLastError", "__hipPushCallConfiguration", "__hipPopCallConfiguration", "hipGetDevice", "hipSetDevice",
             "hipLaunchKernel", "hipModuleLaunchKernel", "hipExtModuleLaunchKernel", "hipGetDeviceCount", "hipStreamIsCapturing")
    for r in ker[a:b]:
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = 0.0 if prev_end is None else (st - prev_end) / 1e3
        calls = by_corr.get(r["Correlation_Id"], [])
        launch = [c for c in calls if "Launch" in c["Function"] or "Memset" in c["Function"] or "Memcpy" in c["Function"]]
        c = (launch or calls or [None])[0]
        lead = (st - int(c["End_Timestamp"])) / 1e3 if c else float("nan")
        tid = c["Thread_Id"][-5:] if c else "?"
        between = ""
        if c and prev_call_end is not None:
            lo, hi = prev_call_end, int(c["Start_Timestamp"])
            names = [x["Function"] + ("@" + x["Thread_Id"][-3:] if x["Thread_Id"] != c["Thread_Id"] else "")
                     for x in api_sorted if lo <= int(x["Start_Timestamp"]) < hi and x["Function"] not in quiet]
            if names:
                between = " | " + ", ".join(names[:14]) + (" ... (%d)" % len(names) if len(names) > 14 else "")
        lines.append("%9.1f %8.1f %7.1f %8.1f  %6s  %s%s" % ((st - t0) / 1e3, (en - st) / 1e3, gap, lead, tid, short(r["Kernel_Name"]), between))
        prev_end = en
        if c:
            prev_call_end = int(c["End_Timestamp"])
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    print(text)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] != "hostseq":
    main(*sys.argv[1:5])


def hostseq(api_path, ker_path, anchor, out=None):
    """Host-order listing of one step: every HIP call (bar the getters) of every thread between the launch calls of the last two
    anchor kernels, with the kernel a launch call started (by correlation id)."""
    api = sorted(csv.DictReader(open(api_path)), key=lambda r: int(r["Start_Timestamp"]))
    ker = {r["Correlation_Id"]: r for r in csv.DictReader(open(ker_path))}
    anchors = [r for r in api if r["Correlation_Id"] in ker and anchor in ker[r["Correlation_Id"]]["Kernel_Name"]]
    lo, hi = int(anchors[-2]["Start_Timestamp"]), int(anchors[-1]["Start_Timestamp"])
    quiet = ("hipGetLastError", "hipPeekAtLastError", "__hipPushCallConfiguration", "__hipPopCallConfiguration", "hipGetDevice", "hipSetDevice",
             "hipGetDeviceCount", "hipStreamIsCapturing", "hipDevicePrimaryCtxGetState")
    lines = ["host_us  dur_us thread  call [-> kernel]"]
    for r in api:
        t = int(r["Start_Timestamp"])
        if not (lo <= t < hi) or r["Function"] in quiet:
            continue
        k = ker.get(r["Correlation_Id"])
        lines.append("%8.1f %6.1f %6s  %s%s" % ((t - lo) / 1e3, (int(r["End_Timestamp"]) - t) / 1e3, r["Thread_Id"][-5:], r["Function"],
                                               (" -> " + short(k["Kernel_Name"])) if k else ""))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    print(text)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "hostseq":
    hostseq(*sys.argv[2:6])

#!/usr/bin/env python3
"""Host API calls and kernel dispatches of one steady-state device step on ONE time axis (rocprofv3 --kernel-trace --hip-trace):
    python3 scripts/host_timeline.py <dir with *_kernel_trace.csv and *_hip_api_trace.csv> [step_from_the_end=5]
Answers: is the device waiting for the host between two steps, and for which call?"""
import csv, glob, os, sys

d = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 5
kf = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
hf = glob.glob(os.path.join(d, "**", "*hip_api_trace.csv"), recursive=True)[0]
def short(n):
    return n.split("(")[0].replace("void ", "").replace("rfs::", "")[:56]
K = [r for r in csv.DictReader(open(kf)) if r["Kind"] == "KERNEL_DISPATCH"]
K.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(K) if short(r["Kernel_Name"]).startswith("k_prep_joint")]
i0, i1 = starts[-back - 1], starts[-back + 1]          # two steps
t0 = int(K[i0]["Start_Timestamp"]); t1 = int(K[i1]["Start_Timestamp"])
ev = []
for r in K[i0:i1]:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "GPU  s%s" % r["Stream_Id"], short(r["Kernel_Name"])))
H = list(csv.DictReader(open(hf)))
for r in H:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e < t0 - 1_000_000 or s > t1:
        continue
    name = r["Function"]
    if name in ("hipGetLastError", "hipSetDevice", "hipGetDevice", "hipPeekAtLastError", "__hipPushCallConfiguration", "__hipPopCallConfiguration",
                "hipGetDeviceCount", "hipEventCreateWithFlags", "hipEventDestroy", "hipCtxGetCurrent"):
        continue
    if e - s < 20_000 and name in ("hipEventRecord", "hipStreamWaitEvent", "hipEventQuery", "hipStreamGetCaptureInfo", "hipStreamIsCapturing"):
        continue                                      # (quick bookkeeping calls: not what anybody waits for)
    ev.append((s, e, "HOST t%s" % r["Thread_Id"][-4:], name))
ev.sort()
for s, e, who, name in ev:
    print(f"{(s - t0) / 1e6:9.3f} {(e - t0) / 1e6:9.3f} {(e - s) / 1e6:8.3f}  {who:12s} {name}")

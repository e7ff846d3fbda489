#!/usr/bin/env python3
"""One steady-state leapfrog step out of a rocprofv3 --kernel-trace of the bench command:
    python3 scripts/step_timeline.py <kernel_trace.csv> [step_index_from_the_end=3] > profiles/<tag>_step_timeline.txt
Prints kernel, start, end, duration (ms, relative to the step's k_prep_joint) and the stream / queue it ran on."""
import csv, sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
def short(n):
    n = n.split("(")[0].replace("void ", "").replace("rfs::", "")
    return n[:64]
rows = [r for r in rows if r["Kind"] == "KERNEL_DISPATCH"]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if short(r["Kernel_Name"]).startswith("k_prep_joint")]      # (the drift rides in it)
i0 = starts[-back - 1]; i1 = starts[-back]
t0 = int(rows[i0]["Start_Timestamp"])
print("# one leapfrog step (8192 chains, configs[1], warm-started root search) inside `rocprofv3 --kernel-trace -- python3 bench.py --gpus 1")
print("# --no-cpu-baseline --headline-only`: kernel, start ms, end ms, duration ms relative to the start of the step's k_prep_joint; stream id")
print("# (stream of the RF sweeps = the caller's; the SWD kernels on the library's second stream, the hand-back search + the listed chains'")
print("# eigenfunctions on a third; all share the chip)")
for r in rows[i0:i1]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    print(f"{short(r['Kernel_Name']):66s}{s:9.3f}{e:9.3f}{e - s:9.3f}   stream {r['Stream_Id']} queue {r['Queue_Id']}")
print(f"# next step's k_prep_joint starts at {(int(rows[i1]['Start_Timestamp']) - t0) / 1e6:.3f} ms")

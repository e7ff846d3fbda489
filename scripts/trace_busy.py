#!/usr/bin/env python3
"""GPU busy fraction and step period out of a rocprofv3 --kernel-trace csv: python3 scripts/trace_busy.py <kernel_trace.csv> [last_n_steps=20]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kind"] == "KERNEL_DISPATCH"]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
pre = [i for i, r in enumerate(rows) if "k_prep_joint" in r["Kernel_Name"]]
i0, i1 = pre[-N - 1], pre[-1]
t0, t1 = int(rows[i0]["Start_Timestamp"]), int(rows[i1]["Start_Timestamp"])
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows[i0:i1])
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"{N} steps: period {(t1 - t0) / N / 1e6:.3f} ms, GPU busy (union of kernel intervals) {busy / N / 1e6:.3f} ms per step = {busy / (t1 - t0):.1%}")
tot = {}
for r in rows[i0:i1]:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("rfs::", "")[:40]
    tot[n] = tot.get(n, 0) + int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for n, v in sorted(tot.items(), key=lambda kv: -kv[1])[:12]:
    print(f"  {n:42s}{v / N / 1e6:8.3f} ms per step")

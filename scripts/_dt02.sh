#!/bin/bash
export TMPDIR=/tmp; root=$PWD; mkdir -p gpurun_out/d2
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/d2/stats -- python3 $root/scripts/_dt02.py "$@" > $root/gpurun_out/d2/out.log 2> $root/gpurun_out/d2/err.log )
cat gpurun_out/d2/out.log
python3 - <<PYEOF
import csv, glob
rows = [r for r in csv.DictReader(open(glob.glob("gpurun_out/d2/stats/*/*kernel_trace.csv")[0])) if r["Kind"] == "KERNEL_DISPATCH"]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.split("(")[0].replace("void ", "").replace("rfs::", "")[:40]
st = [int(r["Start_Timestamp"]) for r in rows if short(r["Kernel_Name"]).startswith("k_prep_joint")]
T0 = st[0]
iv = [(b - a) / 1e6 for a, b in zip(st[:-1], st[1:])]
# second leg = the last 161 steps
n2 = 161
print("steps", len(st), "median interval first leg", sorted(iv[50:450])[200], "second leg", sorted(iv[-150:])[75], "mean second", sum(iv[-150:]) / 150)
tstart2 = st[-n2]
print("second-leg intervals:", [round(v, 1) for v in iv[-150:]][:150])
srch = [((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, (int(r["Start_Timestamp"]) - tstart2) / 1e6, short(r["Kernel_Name"])) for r in rows if "roots" in r["Kernel_Name"] and int(r["Start_Timestamp"]) > tstart2]
print("searches in the second leg: n", len(srch), "total ms", sum(a for a, b, c in srch))
import collections
for name, lo, hi in (("first leg", st[100], st[450]), ("last leg", tstart2, st[-1])):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        t = int(r["Start_Timestamp"])
        if lo <= t < hi:
            a = acc[short(r["Kernel_Name"])]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - t) / 1e6
    nst = sum(1 for v in st if lo <= v < hi)
    print(name, "steps", nst, {k[:26]: (round(v[0] / nst, 2), round(v[1] / nst, 3)) for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1])[:12]})
srch.sort(reverse=True)
print("longest:", [(round(a, 1), round(b), c[:18]) for a, b, c in srch[:12]])
PYEOF
python3 scripts/step_timeline.py $(find gpurun_out/d2/stats -name '*kernel_trace.csv') 30 | grep -v 'at::native\|rocclr' > gpurun_out/d2/tl_slow.txt
python3 scripts/step_timeline.py $(find gpurun_out/d2/stats -name '*kernel_trace.csv') 31 | grep -v 'at::native\|rocclr' >> gpurun_out/d2/tl_slow.txt
rm -rf gpurun_out/d2/stats

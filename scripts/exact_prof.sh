#!/bin/bash
# k_swd_exact / k_swd_warm: duration (serial step) and VALU instructions per launch.  bash scripts/exact_prof.sh <outdir>
out=${1:-gpurun_out/exact_prof}
export TMPDIR=/tmp
mkdir -p $out
RFS_SERIAL=1 timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU --kernel-trace --output-format csv -d $out/g1 -- python3 scripts/prof_run.py 1 4 > $out/g1.log 2>&1
python3 - $out <<PYEOF
import csv, glob, collections, sys
out = sys.argv[1]
f = glob.glob(out + "/g1/*/*counter_collection.csv")
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("rfs::", "")[:40]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k in agg:
    if any(s in k for s in ("k_swd_exact", "k_swd_warm<", "k_swd_warm_check")):
        print(k, {c: "%.3e" % (v / cnt[(k, c)]) for c, v in agg[k].items()})
dur = collections.defaultdict(list)
for r in csv.DictReader(open(glob.glob(out + "/g1/*/*kernel_trace.csv")[0])):
    dur[r["Kernel_Name"].split("(")[0][-40:]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, v in dur.items():
    if "swd_exact" in k or "swd_warm" in k: print(k, "ms %.4f" % (sum(v) / len(v)))
PYEOF
rm -rf $out/g1

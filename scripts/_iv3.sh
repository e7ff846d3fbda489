#!/bin/bash
export TMPDIR=/tmp; root=$PWD; mkdir -p gpurun_out/iv; export WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29531
cd /tmp; timeout 900 rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/iv/stats -- python3 $root/bench.py --gpus 1 --no-cpu-baseline --headline-only --warmup 250 --steps 100 "$@" > $root/gpurun_out/iv/bench.json 2> $root/gpurun_out/iv/err.log; cd $root
python3 - <<PYEOF
import csv, glob
rows = [r for r in csv.DictReader(open(glob.glob("gpurun_out/iv/stats/*/*kernel_trace.csv")[0])) if r["Kind"] == "KERNEL_DISPATCH"]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.split("(")[0].replace("void ", "").replace("rfs::", "")[:40]
st = [int(r["Start_Timestamp"]) for r in rows if short(r["Kernel_Name"]).startswith("k_prep_joint")]
iv = [(b - a) / 1e6 for a, b in zip(st[:-1], st[1:])]
T0 = st[0]
big = sorted(range(len(iv)), key=lambda i: -iv[i])[:4]
print("longest step intervals:", [(i, round(iv[i], 1), round((st[i] - T0) / 1e6)) for i in big])
srch = [((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, (int(r["Start_Timestamp"]) - T0) / 1e6, short(r["Kernel_Name"])) for r in rows if "roots" in r["Kernel_Name"]]
srch.sort(reverse=True)
print("longest searches (ms, start ms, kernel):", [(round(a, 1), round(b), c[:22]) for a, b, c in srch[:8]])
others = sorted([((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, (int(r["Start_Timestamp"]) - T0) / 1e6, short(r["Kernel_Name"])) for r in rows if "roots" not in r["Kernel_Name"]], reverse=True)[:6]
print("longest other kernels:", [(round(a, 1), round(b), c[:28]) for a, b, c in others])
PYEOF
rm -rf gpurun_out/iv/stats

#!/bin/bash
# experiment: run-up 1 instead of 2 in the reference-root stage (identity vs the full search, step time)
mkdir -p gpurun_out/ru
for s in "3 1" "4 1"; do timeout 300 python3 scripts/exact_gpu.py 8192 8 0.02 $s > gpurun_out/ru/exact_${s// /_}.log 2>&1; done
for s in "3 2" "3 1" "4 1" "3 2"; do set -- $s
RFS_EXACT_GROUP=$1 RFS_EXACT_RUNUP=$2 timeout 300 python3 bench.py --gpus 1 --no-cpu-baseline --headline-only --warmup 250 --steps 150 > gpurun_out/ru/bench_$1_$2.json 2> gpurun_out/ru/bench_$1_$2.err
python3 - <<PY
import json
b=json.loads(open("gpurun_out/ru/bench_$1_$2.json").read().strip().splitlines()[-1])
print("G $1 RU $2:", round(b["ms_per_step"],3), "ms", round(b["accept_ratio"],3), b["root_search"])
PY
done
tail -8 gpurun_out/ru/exact_*.log

#!/bin/bash
mkdir -p gpurun_out/g
for s in "4 2" "5 2" "6 2" "4 2" "5 2" "6 2"; do set -- $s
RFS_EXACT_GROUP=$1 RFS_EXACT_RUNUP=$2 timeout 300 python3 bench.py --gpus 1 --no-cpu-baseline --headline-only --warmup 250 --steps 150 > gpurun_out/g/bench_$1_$2.json 2> gpurun_out/g/bench_$1_$2.err
python3 - <<PY
import json
b=json.loads(open("gpurun_out/g/bench_$1_$2.json").read().strip().splitlines()[-1])
print("G $1 RU $2:", round(b["ms_per_step"],3), "ms", round(b["accept_ratio"],3), round(b["root_search"]["secular_evals_per_item_reference_root_stage"],2), round(b["root_search"]["chains_handed_back_to_the_full_search_per_step"],1), b["kernel_ms_per_step"]["swd_exact"])
PY
done

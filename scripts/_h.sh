#!/bin/bash
mkdir -p gpurun_out/h
timeout 900 python3 -m pytest tests/test_gpu_warm.py -x -q -m gpu --timeout 300 -k "dense_grid or option_zero or large_steps or love_group" 2>&1 | tail -3
for i in 1 2 3; do
timeout 300 python3 bench.py --gpus 1 --no-cpu-baseline --headline-only --warmup 250 --steps 150 "$@" > gpurun_out/h/bench_$i.json 2> gpurun_out/h/bench_$i.err
python3 - <<PY
import json
b=json.loads(open("gpurun_out/h/bench_$i.json").read().strip().splitlines()[-1])
r=b["root_search"]
print("run $i:", round(b["ms_per_step"],3), "ms", round(b["accept_ratio"],3), round(r["secular_evals_per_item_warm_start_and_branch_test"],2), round(r["secular_evals_per_item_reference_root_stage"],2), round(r["chains_handed_back_to_the_full_search_per_step"],1), {k: round(v,2) for k,v in b["kernel_ms_per_step"].items()})
PY
done

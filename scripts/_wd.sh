#!/bin/bash
mkdir -p gpurun_out/wd
timeout 600 python3 -m pytest tests/test_gpu_warm.py -x -q -m gpu --timeout 300 -s -k "dense_grid or option_zero or large_steps" > gpurun_out/wd/tests.log 2>&1; echo "tests rc $?"
tail -15 gpurun_out/wd/tests.log
for v in 1 0 1 0; do
RFS_WALK_DENSE=$v timeout 300 python3 bench.py --gpus 1 --no-cpu-baseline --headline-only --warmup 250 --steps 150 > gpurun_out/wd/bench_$v.json 2> gpurun_out/wd/bench_$v.err
python3 - <<PY
import json
b=json.loads(open("gpurun_out/wd/bench_$v.json").read().strip().splitlines()[-1])
print("dense $v:", round(b["ms_per_step"],3), "ms", round(b["accept_ratio"],3), b["root_search"], b["kernel_ms_per_step"])
PY
done

#!/bin/bash
# A/B of an environment knob on the headline: bash scripts/_ab.sh VAR val1 val2 ...
mkdir -p gpurun_out/ab; var=$1; shift
for rep in 1 2; do for v in "$@"; do
env $var=$v timeout 300 python3 bench.py --gpus 1 --no-cpu-baseline --headline-only --warmup 250 --steps 150 > gpurun_out/ab/b.json 2> gpurun_out/ab/b.err
python3 - <<PY
import json
b=json.loads(open("gpurun_out/ab/b.json").read().strip().splitlines()[-1])
print("$var=$v:", round(b["ms_per_step"],3), "ms", round(b["accept_ratio"],3), round(b["root_search"]["chains_handed_back_to_the_full_search_per_step"],1))
PY
done; done

#!/bin/bash
# one step's kernel timeline of the headline command (rocprofv3 --kernel-trace): gpurun_out/tl/timeline.txt
export TMPDIR=/tmp; root=$PWD; mkdir -p gpurun_out/tl; export WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29531
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/tl/stats -- python3 $root/bench.py --gpus 1 --no-cpu-baseline --headline-only --warmup 250 --steps 100 "$@" > $root/gpurun_out/tl/bench.json 2> $root/gpurun_out/tl/err.log )
python3 scripts/step_timeline.py $(find gpurun_out/tl/stats -name "*kernel_trace.csv") 5 > gpurun_out/tl/timeline.txt
rm -rf gpurun_out/tl/stats
grep -v "at::native\|rocclr" gpurun_out/tl/timeline.txt

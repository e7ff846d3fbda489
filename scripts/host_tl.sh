#!/bin/bash
# host + device timeline of two steps of the headline command: gpurun_out/htl/timeline.txt
export TMPDIR=/tmp; root=$PWD; mkdir -p gpurun_out/htl; export WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29531
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --hip-trace --output-format csv -d $root/gpurun_out/htl/stats -- python3 $root/bench.py --gpus 1 --no-cpu-baseline --headline-only --warmup 250 --steps 60 "$@" > $root/gpurun_out/htl/bench.json 2> $root/gpurun_out/htl/err.log )
python3 scripts/host_timeline.py gpurun_out/htl/stats 5 > gpurun_out/htl/timeline.txt
rm -rf gpurun_out/htl/stats
wc -l gpurun_out/htl/timeline.txt

#!/bin/bash
# Two half-populations side by side on ONE GPU (two processes, 4096 chains each) against one population of 8192:
# what would a two-half software pipeline of the step be worth?  (RFS_BENCH_SHARED_GPU: functional N>1 path, all ranks on GPU 0)
mkdir -p gpurun_out/hp
A="--steps 150 --warmup 300 --no-cpu-baseline --headline-only"
python3 bench.py --gpus 1 $A > gpurun_out/hp/one_8192.json 2> gpurun_out/hp/one_8192.err
python3 bench.py --gpus 1 --chains 4096 $A > gpurun_out/hp/one_4096.json 2> gpurun_out/hp/one_4096.err
RFS_BENCH_SHARED_GPU=1 python3 bench.py --gpus 2 --chains 4096 $A > gpurun_out/hp/two_4096.json 2> gpurun_out/hp/two_4096.err
GPU_MAX_HW_QUEUES=8 RFS_BENCH_SHARED_GPU=1 python3 bench.py --gpus 2 --chains 4096 $A > gpurun_out/hp/two_4096_q8.json 2> gpurun_out/hp/two_4096_q8.err
GPU_MAX_HW_QUEUES=8 python3 bench.py --gpus 1 $A > gpurun_out/hp/one_8192_q8.json 2> gpurun_out/hp/one_8192_q8.err
RFS_BENCH_SHARED_GPU=1 python3 bench.py --gpus 4 --chains 2048 $A > gpurun_out/hp/four_2048.json 2> gpurun_out/hp/four_2048.err
for f in one_8192 one_4096 two_4096 two_4096_q8 one_8192_q8 four_2048; do
  python3 -c "
import json,sys
try:
    d=json.loads(open('gpurun_out/hp/$f.json').read().strip().splitlines()[-1])
    print('$f', 'value', d['value'], 'ms/step', d['ms_per_step'], 'accept', d.get('accept_ratio'))
except Exception as e:
    print('$f', 'failed', e)
"
done

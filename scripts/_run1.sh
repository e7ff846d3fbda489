export TMPDIR=/tmp
mkdir -p gpurun_out
AB_STEPS=200 bash scripts/ab_bench.sh 2 "-" "RFS_BG_STREAM_PRIORITY=1" > gpurun_out/r06_ab_bgprio.txt 2>&1; cat gpurun_out/r06_ab_bgprio.txt
for v in 0 1 0 1; do RFS_BG_STREAM_PRIORITY=$v python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --headline-only 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bgprio $v K=20: ms/step %.3f sustained %.3f' % (d['ms_per_step'], d.get('sustained_ms_per_step', 0)))"; done

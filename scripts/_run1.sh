export TMPDIR=/tmp
timeout 300 python3 scripts/host_cpu_threads.py 2>&1 | tail -5
for v in 1 0; do RFS_HOST_SPIN=$v timeout 300 python3 bench.py --headline-only --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('spin $v', d['ms_per_step'], d['value'], 'host cpu ms/step', d.get('host_cpu_ms_per_step'))"; done

#!/bin/bash
# Same-box A/B of the headline: alternating bench.py runs, each variant = "ENV=VAL,ENV=VAL" (use - for none).
#   bash scripts/ab_bench.sh reps "RFSURF_LIB=$PWD/ab/librfsurf_r04.so,RFS_FLOW_LEGACY_TAIL=1" "-"
reps=$1; shift
mkdir -p gpurun_out/ab
for r in $(seq 1 $reps); do
  i=0
  for v in "$@"; do
    i=$((i+1))
    envs=""; [ "$v" != "-" ] && envs=$(echo "$v" | tr ',' ' ')
    out=$(env $envs python3 bench.py --steps ${AB_STEPS:-200} --warmup 300 --no-cpu-baseline --headline-only ${AB_ARGS} 2>gpurun_out/ab/err_$i.log | tail -1)
    python3 -c "
import json,sys
d=json.loads(sys.argv[1]); print('variant $i [$v]: ms/step %.3f  value %.0f  accept %.3f  handed back %.1f  warm %.2f exact %.2f' % (d['ms_per_step'], d['value'], d['accept_ratio'], d.get('handed_back_per_step', -1), d.get('evals_per_item_warm', -1), d.get('evals_per_item_exact', -1)))" "$out" 2>/dev/null || { echo "variant $i failed"; tail -3 gpurun_out/ab/err_$i.log; }
  done
done

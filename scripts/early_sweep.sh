#!/bin/bash
# sweep of the number of early eigenfunction periods (config 2)
for k in ${KS:-off 0 8 12 16 18 20 24}; do
  if [ $k = off ]; then export RFS_NO_EARLY_EIGEN=1; unset RFS_EARLY_EIGEN_K; else unset RFS_NO_EARLY_EIGEN; export RFS_EARLY_EIGEN_K=$k; fi
  echo -n "K=$k "; timeout 100 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernel_ms_per_launch'].items()})"
done
unset RFS_NO_EARLY_EIGEN RFS_EARLY_EIGEN_K
echo -n "auto "; timeout 100 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3))"

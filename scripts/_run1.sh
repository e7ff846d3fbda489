export TMPDIR=/tmp
mkdir -p gpurun_out
NMAX=8192 timeout 3000 python3 scripts/flow_parity_stats.py 240 320 400 480 560 640 720 800 > gpurun_out/r06_parity_stats_final.txt 2>&1
grep -E "^step|TOTAL" gpurun_out/r06_parity_stats_final.txt | sed 's/misfit_p99.*grad_max /grad_max /' | cut -c1-420

export TMPDIR=/tmp
mkdir -p gpurun_out
bash scripts/ab_bench.sh 2 "-" "RFS_OPTS=rf_hold=1" "RFS_OPTS=rf_hold=2" "RFS_OPTS=rf_hold=3" > gpurun_out/r06_ab_hold.txt 2>&1; cat gpurun_out/r06_ab_hold.txt

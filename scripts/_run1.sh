export TMPDIR=/tmp
mkdir -p gpurun_out
RFSURF_LIB=$PWD/ab/librfsurf_dbg.so timeout 200 python3 scripts/coop_debug.py 2>&1 | tail -2
timeout 600 python3 -m pytest tests/test_gpu_warm.py -x -q -m gpu -k "16_lanes or in_rounds" > gpurun_out/r06_coop_test.txt 2>&1; tail -5 gpurun_out/r06_coop_test.txt
for o in "swd_exact_coop=0" "swd_exact_coop=1" ; do RFS_OPTS=$o timeout 100 python3 scripts/config0_flow.py 300 1 2>&1 | grep "chain(s)"; done

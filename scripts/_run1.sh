export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1700 python3 -m pytest tests -x -q -m gpu > gpurun_out/r06_gpu_suite.txt 2>&1; tail -6 gpurun_out/r06_gpu_suite.txt
bash scripts/ab_bench.sh 2 "RFSURF_LIB=$PWD/ab/librfsurf_base6.so,RFS_FLOW_RECORDS=0" "-" > gpurun_out/r06_ab_all.txt 2>&1
cat gpurun_out/r06_ab_all.txt

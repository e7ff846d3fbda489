export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests/test_gpu_samplers.py tests/test_gpu_resume.py tests/test_gpu_multirank.py tests/test_gpu_warm.py -x -q -m gpu > gpurun_out/r06_host_tests.txt 2>&1; tail -4 gpurun_out/r06_host_tests.txt
bash scripts/ab_bench.sh 2 "RFS_FLOW_RECORDS=0" "-" > gpurun_out/r06_ab_records.txt 2>&1
cat gpurun_out/r06_ab_records.txt

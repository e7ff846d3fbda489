export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1700 python3 -m pytest tests -q -m gpu > gpurun_out/r06_gpu_suite.txt 2>&1; tail -3 gpurun_out/r06_gpu_suite.txt

export TMPDIR=/tmp
mkdir -p gpurun_out
bash scripts/ab_bench.sh 2 "RFS_OPTS=rf_store_hyp=0" "-" > gpurun_out/r06_ab_hst.txt 2>&1; cat gpurun_out/r06_ab_hst.txt
timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_fullsize.py tests/test_gpu_fullsize_configs.py -x -q -m gpu > gpurun_out/r06_rf_tests.txt 2>&1; tail -4 gpurun_out/r06_rf_tests.txt

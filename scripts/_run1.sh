export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1700 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_driver.json 2> gpurun_out/r06_bench_driver.err ) 2>&1 | tail -3
tail -c 2800 gpurun_out/r06_bench_driver.json | head -c 1200; echo; wc -c gpurun_out/r06_bench_driver.json
python3 __graft_entry__.py > /dev/null 2>&1; python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1

export TMPDIR=/tmp
mkdir -p gpurun_out
bash scripts/ab_bench.sh 2 "RFS_OPTS=swd_exact_overlap=0" "RFS_WALK_STREAM_PRIORITY=1" "RFS_WALK_STREAM_PRIORITY=1,RFS_OPTS=swd_exact_overlap=0" > gpurun_out/r06_ab_overlap.txt 2>&1; cat gpurun_out/r06_ab_overlap.txt
RFS_WALK_STREAM_PRIORITY=1 timeout 300 python3 scripts/event_timeline.py 100 2>&1 | tail -12

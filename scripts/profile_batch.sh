#!/bin/bash
# Profile batch of a round (run on the GPU box from the repo root):  bash scripts/profile_batch.sh r02
#   gpurun_out/<tag>_bench.json      the bench line (with cpu_baseline)
#   gpurun_out/<tag>_stats/          rocprofv3 --kernel-trace --stats of the bench command (the rank process itself:
#                                    WORLD_SIZE=1 in the environment makes bench.py run as rank 0 without spawning a child --
#                                    a process that the profiler's library has attached to the GPU must not start another)
#   gpurun_out/<tag>_pmc/<group>/    one rocprofv3 --pmc pass per counter group (scripts/prof_run.py 8192 3)
# Afterwards (anywhere):  python3 scripts/pmc_summary.py <tag> gpurun_out/<tag>_pmc ; copy the stats csv to profiles/.
tag=${1:-r02}
export TMPDIR=/tmp
root=$PWD
mkdir -p gpurun_out
timeout 600 python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
export WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29531
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/${tag}_stats -- python3 $root/bench.py --gpus 1 --no-cpu-baseline > $root/gpurun_out/${tag}_stats_bench.json 2> /dev/null )
unset WORLD_SIZE RANK LOCAL_RANK
i=0
for g in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_INSTS_LDS"; do
  i=$((i+1))
  ( cd $root && timeout 300 rocprofv3 --pmc $g --kernel-trace --output-format csv -d gpurun_out/${tag}_pmc/g$i -- python3 scripts/prof_run.py 8192 3 > gpurun_out/${tag}_pmc_g$i.log 2>&1 )
done
ls gpurun_out/${tag}_stats/* gpurun_out/${tag}_pmc/* | head -30

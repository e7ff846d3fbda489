#!/bin/bash
# Profile batch of a round (run on the GPU box from the repo root):  bash scripts/profile_batch.sh r05
#   gpurun_out/<tag>_bench.json      the bench line (with cpu_baseline)
#   gpurun_out/<tag>_stats/          rocprofv3 --kernel-trace --stats of the bench command (the rank process itself:
#                                    WORLD_SIZE=1 in the environment makes bench.py run as rank 0 without spawning a child --
#                                    a process that the profiler's library has attached to the GPU must not start another)
#   gpurun_out/<tag>_pmc/c<config>/g<i>/  one rocprofv3 --pmc pass per counter group and configuration (scripts/prof_run.py)
# Afterwards (anywhere):  python3 scripts/pmc_summary.py <tag> <config> 0 gpurun_out/<tag>_pmc/c<config> ; copy the stats csv to profiles/.
tag=${1:-r06}
export TMPDIR=/tmp
root=$PWD
mkdir -p gpurun_out
timeout 1500 python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
cp bench_detail.json gpurun_out/${tag}_bench_detail.json        # (the profiled run below writes its own)
export WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29531
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/${tag}_stats -- python3 $root/bench.py --gpus 1 --no-cpu-baseline --headline-only > $root/gpurun_out/${tag}_stats_bench.json 2> /dev/null )
unset WORLD_SIZE RANK LOCAL_RANK
python3 scripts/step_timeline.py $(find gpurun_out/${tag}_stats -name "*kernel_trace.csv") 5 > gpurun_out/${tag}_step_timeline.txt
cp $(find gpurun_out/${tag}_stats -name "*kernel_stats.csv") gpurun_out/${tag}_bench_kernel_stats.csv
timeout 300 python3 scripts/event_timeline.py 100 > gpurun_out/${tag}_step_timeline_events.txt 2>&1
timeout 300 python3 scripts/host_cpu_threads.py > gpurun_out/${tag}_host_cpu_threads.txt 2>&1
rm -f $(find gpurun_out/${tag}_stats -name "*kernel_trace.csv")
for cfg in 1 3 4; do
  timeout 600 python3 scripts/prof_run.py $cfg 10 hmc > gpurun_out/${tag}_burn_c$cfg.log 2>&1      # burn-in, kept in gpurun_out/burned_c<cfg>.npy
  i=0
  for g in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64"; do
    i=$((i+1))
    ( cd $root && timeout 300 rocprofv3 --pmc $g --kernel-trace --output-format csv -d gpurun_out/${tag}_pmc/c$cfg/g$i -- python3 scripts/prof_run.py $cfg 40 hmc > gpurun_out/${tag}_pmc_c${cfg}_g$i.log 2>&1 )
  done
  python3 scripts/pmc_summary.py $tag $cfg 0 gpurun_out/${tag}_pmc/c$cfg > gpurun_out/${tag}_pmc_summary_c$cfg.txt 2>&1
  rm -rf gpurun_out/${tag}_pmc/c$cfg
done
rm -f gpurun_out/burned_c*.npy
cp profiles/${tag}_pmc_config*.csv profiles/${tag}_counters.json gpurun_out/ 2>/dev/null
ls gpurun_out/${tag}_stats/* | head

#!/bin/bash
# Round 5: what an instruction costs (scripts/instr_cost.hip) and two ways to shorten the reference-root stage's dependent chain:
# fewer periods per lane (swd_exact_group) and the Estrin-form exponential (ab/librfsurf_shortexp.so, -DRFS_EXACT_SHORT_EXP).
mkdir -p gpurun_out
hipcc -O3 --offload-arch=gfx950 scripts/instr_cost.hip -o /tmp/instr_cost 2>/dev/null && timeout 120 /tmp/instr_cost > gpurun_out/instr_cost.txt 2>&1
cat gpurun_out/instr_cost.txt
AB_STEPS=200 timeout 900 bash scripts/ab_bench.sh ${1:-2} "-" "RFS_OPTS=swd_exact_group=3" "RFS_OPTS=swd_exact_group=2" "RFSURF_LIB=$PWD/ab/librfsurf_shortexp.so" 2>&1 | tee gpurun_out/ilp_probe_ab.txt

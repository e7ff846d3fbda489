export TMPDIR=/tmp
mkdir -p gpurun_out
( for a in 8400 8700 9000 9300; do timeout 600 python3 scripts/warm_fuzz_soak.py $a $((a+300)) 2>&1 | tail -1; done ) > gpurun_out/r06_soak_final.txt 2>&1; cat gpurun_out/r06_soak_final.txt | cut -c1-330
timeout 600 python3 scripts/handback_causes.py > gpurun_out/r06_handback_causes.txt 2>&1; tail -4 gpurun_out/r06_handback_causes.txt | cut -c1-900

#!/bin/bash
# The two tests next to which one suite run stopped (EXPERIMENTS.md, "One run of the GPU suite stopped making progress"), N times each in fresh processes, with the
# watchdog thread at 90 s: a hang prints every thread's stack.
mkdir -p gpurun_out
n=0
for i in $(seq 1 ${1:-30}); do
  timeout 200 python3 -X faulthandler -m pytest -q --timeout 90 --timeout_method=thread \
     tests/test_gpu_rf_time.py::test_time_domain_two_layers_and_many_layers \
     tests/test_gpu_rf_time.py::test_reference_smoke_script_configuration \
     tests/test_gpu_samplers.py::test_hmc_reproduces_reference_ranks_0_and_1 > gpurun_out/hang_hunt_last.log 2>&1
  rc=$?
  if [ $rc -ne 0 ]; then n=$((n+1)); cp gpurun_out/hang_hunt_last.log gpurun_out/hang_hunt_fail_$i.log; echo "run $i: rc $rc"; fi
done
echo "hang hunt: $n failures of ${1:-30} runs; last: $(tail -1 gpurun_out/hang_hunt_last.log)"

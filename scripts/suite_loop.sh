#!/bin/bash
# The whole -m gpu suite N times on one box, each run bounded, with the slowest tests and (on a hang) every thread's stack.
mkdir -p gpurun_out
for i in $(seq 1 ${1:-3}); do
  timeout 900 python3 -X faulthandler -m pytest tests -q -m gpu -x --timeout 240 --timeout_method=thread --durations=12 > gpurun_out/suite_loop_$i.log 2>&1
  echo "run $i: rc $? $(tail -1 gpurun_out/suite_loop_$i.log)"
done

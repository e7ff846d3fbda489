#!/bin/bash
# GPU clock / power while the bench workload runs (read-only rocm-smi queries): is the step power-limited?
python scripts/prof_run.py 8192 5000 > /dev/null 2>&1 &
PID=$!
sleep 20
for i in 1 2 3 4; do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|Power \(W\)|junction" | head -4; sleep 2; done
kill $PID 2>/dev/null; wait $PID 2>/dev/null
rocm-smi --showmaxpower 2>/dev/null | grep -i "power (w)" | head -2

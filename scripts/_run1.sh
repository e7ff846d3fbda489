export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_warm.py -q -m gpu -k "16_lanes" 2>&1 | tail -2

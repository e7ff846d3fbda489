#!/bin/bash
# kernel trace of ONE configs[0] chain (scripts/config0_flow.py) -> gpurun_out/c0_trace.csv.gz; optional context options in $1
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
RFS_CTX_OPTS="$1" rocprofv3 --kernel-trace --output-format csv -d gpurun_out/c0 -- python3 scripts/config0_flow.py 100 1 2>&1 | tail -2
f=$(ls gpurun_out/c0/*/*kernel_trace.csv | head -1); gzip -c $f > gpurun_out/c0_trace.csv.gz; rm -rf gpurun_out/c0

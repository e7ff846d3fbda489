#!/bin/bash
# A/B builds of the library: product (rfsurfhmc_amd/librfsurf_hip.so) and the profiling variant ab/librfsurf_prof.so
set -e
cd "$(dirname "$0")/.."
mkdir -p ab
python -c "from rfsurfhmc_amd import build; build.build(force=True)" 2>&1 | grep -E "error" || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -Wno-pass-failed -DRFS_COOP_PROFILE $EXTRA \
    rfsurfhmc_amd/csrc/rfsurf_hip.hip -o ab/librfsurf_prof.so -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib 2>&1 | grep -E "error" || true
ls -la rfsurfhmc_amd/librfsurf_hip.so ab/librfsurf_prof.so

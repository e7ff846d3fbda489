#!/bin/bash
# Rebuild the product library (rfsurfhmc_amd/librfsurf_hip.so).  A/B comparisons: scripts/ab_libs.py (two builds, one box).
set -e
cd "$(dirname "$0")/.."
python -c "from rfsurfhmc_amd import build; build.build(force=True)" 2>&1 | grep -E "error" || true
ls -la rfsurfhmc_amd/librfsurf_hip.so

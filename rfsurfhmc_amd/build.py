"""Build librfsurf_hip.so for gfx950 with hipcc (in-tree, so the .so travels with gpurun)."""
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librfsurf_hip.so")
SOURCES = ["rfsurf_hip.hip"]


def _inputs():
    """Everything the library is compiled from: the translation unit, every header beside it, the public header."""
    return ([os.path.join(CSRC, f) for f in SOURCES] + sorted(glob.glob(os.path.join(CSRC, "*.hpp")))
            + [os.path.join(HERE, "..", "include", "rfsurf.h")])


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(f) > t for f in _inputs())


RNGLIB = os.path.join(HERE, "librngbatch.so")
RNGSRC = os.path.join(HERE, "csrc_host", "rngbatch.c")


def build_host_helpers(force: bool = False) -> str:
    """gcc -> rfsurfhmc_amd/librngbatch.so: the batched samplers' C loop over per-chain numpy legacy RNG streams."""
    if force or not os.path.exists(RNGLIB) or os.path.getmtime(RNGSRC) > os.path.getmtime(RNGLIB):
        subprocess.run(["gcc", "-O2", "-fPIC", "-shared", RNGSRC, "-o", RNGLIB, "-lm"], check=True)
    return RNGLIB


def build(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 ... -> rfsurfhmc_amd/librfsurf_hip.so"""
    try:                                   # optional: without it the samplers draw chain by chain in Python
        build_host_helpers(force)
    except (FileNotFoundError, subprocess.CalledProcessError) as e:
        print(f"rfsurfhmc_amd.build: librngbatch.so not built ({e}); the samplers use their Python RNG path")
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
           os.path.join(CSRC, "rfsurf_hip.hip"), "-o", LIB, "-L/opt/rocm/lib", "-lrocfft",
           "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=False))

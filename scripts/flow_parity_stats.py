"""tests/test_gpu_flow_parity.py's configs[1] comparison at several device steps: how often the measured mode leaves the 1e-5
contract (misfit of completed trajectories, gradient of mid-trajectory chains, against the oracle).  RFS_OPTS selects options.
    python3 scripts/flow_parity_stats.py 150 250 300 400"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench
import test_gpu_flow_parity as T
from oracle import oracle as O
from rfsurfhmc_amd.pyhmc.hmc import HamitonianMC


def main():
    O.build(ref=False)
    n, nt, nchain = 30, 512, 8192
    tot = dict(n_end=0, n_mid=0, m_over=0, g_over=0, m_max=0.0, g_max=0.0, same=0, other=0, g_over_same=0, g_over_other=0, g_max_same=0.0)
    for s0 in [int(a) for a in sys.argv[1:]] or [200]:
        joint, t = T._joint(n, nt, 0.1)
        bounds = bench.bounds_of(bench.true_model(n))
        smp = HamitonianMC(joint, bounds, bench.TUNED_DT, [5, 20], 10, 991206, 140, 20, myrank=0, name="parity", outdir=None,
                           nchains=nchain, verbose=False, store_syn=False)
        b, a = T._capture(smp, bench.make_models(nchain, 991206, n), s0, roots_of=lambda: joint._ensure(n).last_roots(nchain))
        rfpar = (bench.RAY_P, nt, 0.1, bench.GAUSS, bench.TSHIFT, bench.WATER, "P", "freq")
        r = T._against_the_oracle(b, a, bounds, joint, t, rfpar, nt, int(os.environ.get("NMAX", 512)), f"step {s0} [{os.environ.get('RFS_OPTS', 'default')}]")
        tot["n_end"] += r["n_end"]; tot["n_mid"] += r["n_mid"]
        tot["m_over"] += round(r["misfit_share_above_1e5"] * r["n_end"]); tot["g_over"] += round(r["grad_share_above_1e5"] * r["n_mid"])
        tot["m_max"] = max(tot["m_max"], r["misfit_max"]); tot["g_max"] = max(tot["g_max"], r["grad_max"])
        tot["same"] += r["n_mid_same_roots"]; tot["other"] += r["n_mid_other_root"]
        tot["g_over_same"] += r["grad_above_1e5_same_roots"]; tot["g_over_other"] += r["grad_above_1e5_other_root"]
        tot["g_max_same"] = max(tot["g_max_same"], r["grad_max_same_roots"])
        joint._ctx.close(); joint._ctx = None
    print("TOTAL", os.environ.get("RFS_OPTS", "default"), tot)


if __name__ == "__main__":          # (the oracle pool SPAWNS its workers: they import this file)
    main()

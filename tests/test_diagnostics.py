"""Convergence diagnostics (rfsurfhmc_amd/pyhmc/diagnostics.py) on synthetic chains with known answers."""
import numpy as np

from rfsurfhmc_amd.pyhmc import diagnostics as D


def _ar1(rng, nc, ns, phi, mean=0.0):
    x = np.zeros((nc, ns))
    x[:, 0] = rng.standard_normal(nc) / np.sqrt(1 - phi * phi)
    e = rng.standard_normal((nc, ns))
    for t in range(1, ns):
        x[:, t] = phi * x[:, t - 1] + e[:, t]
    return x + mean


def test_rhat_and_ess_on_ar1_chains():
    rng = np.random.default_rng(0)
    nc, ns = 64, 400
    iid = rng.standard_normal((nc, ns))
    ar = _ar1(rng, nc, ns, 0.8)
    x = np.stack((iid, ar), axis=2)
    r = D.split_rhat(x)
    assert abs(r[0] - 1.0) < 0.01 and abs(r[1] - 1.0) < 0.05   # 200-sample halves of an AR(0.8) chain hold ~22 effective draws
    e = D.ess(x)
    n = nc * ns
    assert 0.8 * n < e[0] < 1.25 * n                      # independent draws: ESS ~ N
    expect = n * (1 - 0.8) / (1 + 0.8)                    # AR(1): N (1 - phi) / (1 + phi)
    assert 0.75 * expect < e[1] < 1.3 * expect
    s = D.summarize(x)
    assert abs(s["mean"][0]) < 0.05 and abs(s["sd"][0] - 1.0) < 0.05
    assert abs(s["q0.5"][1]) < 0.2 and s["q0.025"][0] < -1.8 and s["q0.975"][0] > 1.8


def test_rhat_flags_chains_that_disagree():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((16, 200))
    x[:8] += 3.0                                          # half of the chains sit in another mode
    assert D.split_rhat(x)[0] > 1.5
    y = rng.standard_normal((16, 200)) + np.linspace(0, 4, 200)[None, :]      # all chains still drifting
    assert D.split_rhat(y)[0] > 1.3
    assert D.split_rhat(rng.standard_normal((16, 200)))[0] < 1.02

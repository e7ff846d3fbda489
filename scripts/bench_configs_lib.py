import numpy as np


def models(n, thk0, vs0, nchain, seed=1):
    """Sorted-prior random models around (vs0, thk0), as scripts/bench_configs.py draws them."""
    lo = np.maximum(vs0 - 0.8 * vs0, 1.5); hi = np.minimum(vs0 + 0.8 * vs0, 5.0)
    rng = np.random.default_rng(seed)
    v = np.sort(lo + (hi - lo) * rng.random((nchain, n)), axis=1)
    h = thk0 * (0.8 + 0.4 * rng.random((nchain, n))); h[:, -1] = 1.0
    return np.hstack((v, h))

import sys, numpy as np
sys.path.insert(0, '.')
from oracle import oracle as orc
orc.build(ref=False)
from rfsurfhmc_amd.model.lib import librf
thk = np.array([6., 6, 13., 5, 10, 30, 0]); vs = np.array([3.2, 2.8, 3.46, 3.3, 3.9, 4.5, 4.7])
vp, rho, _, _ = orc.empirical_relation(vs); q = np.full(7, 9999.)
rf0, kl0 = orc.librf.kernel_all(thk, rho, vp, vs, q, q, 0.045, 125, 0.4, 1.5, 5.0, "time", 0.001, "P")
rf1, kl1 = librf.kernel_all(thk, rho, vp, vs, q, q, 0.045, 125, 0.4, 1.5, 5.0, "time", 0.001, "P")
for ip in range(4):
    for j in range(7):
        a, b = kl1[ip, j], kl0[ip, j]
        e = np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)
        print(ip, j, "%.3e" % e, "%.3e" % np.abs(b).max())

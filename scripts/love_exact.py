"""How many phase velocities of the Love / sphere fixtures (compiled reference) the device reproduces exactly."""
import sys; sys.path.insert(0, '.')
import numpy as np
from oracle import oracle as orc
from rfsurfhmc_amd.model.lib import libsurf
g = np.load('tests/golden/swd_love_sphere_reference.npz')
tot = {}; bad = {}
for name in sorted({k.split("/")[0] for k in g.files if k.endswith("/thk")}):
    thk, vs, t = g[f"{name}/thk"], g[f"{name}/vs"], g[f"{name}/t"]
    vp, rho, _, _ = orc.empirical_relation(vs)
    for wt in ("Rc", "Lc"):
        for sph in (0, 1):
            key = f"{name}/{wt}/{sph}"
            if f"{key}/fwd_c" not in g.files: continue
            c, flag = libsurf.forward(thk, vp, vs, rho, t, wt, 0, bool(sph))
            if not flag or not bool(g[f"{key}/fwd_flag"]): continue
            ref = g[f"{key}/fwd_c"]
            k = (wt, sph)
            tot[k] = tot.get(k, 0) + len(ref); nb = int((c != ref).sum()); bad[k] = bad.get(k, 0) + nb
            if nb: print(key, nb, "of", len(ref), "max rel", float(np.abs(c - ref).max() / ref.max()))
print({k: (bad[k], tot[k]) for k in tot})

"""CPU simulation of the warm-started root refinement (tests/hostsim build of the device math) along leapfrog-like
trajectories of the bench's models, against the bit-exact reference-semantics search at every step.
    python scripts/warm_sim.py [nchain] [nsteps] [dt]
"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from oracle import oracle as O

so = os.path.join(ROOT, "tests", "hostsim", "libhostsim_swd.so")
H = ctypes.CDLL(so)
H.hs_sregn96.restype = ctypes.c_double
DP = ctypes.POINTER(ctypes.c_double); FP = ctypes.POINTER(ctypes.c_float); IP = ctypes.POINTER(ctypes.c_int)
P = lambda a: a.ctypes.data_as(DP); F = lambda a: a.ctypes.data_as(FP); I = lambda a: a.ctypes.data_as(IP)

nchain = int(sys.argv[1]) if len(sys.argv) > 1 else 16
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dt = float(sys.argv[3]) if len(sys.argv) > 3 else 0.002
n = 30
t = np.ascontiguousarray(np.linspace(5, 44, 40)); nt = len(t)
xs = bench.make_models(nchain, seed=991206, n=n)
rng = np.random.default_rng(5)

def f32model(x):
    vs, thk = x[:n], x[n:]
    vp, rho, dadb, drda = O.empirical_relation(vs)
    return [np.ascontiguousarray(np.asarray(v, dtype=np.float64).astype(np.float32)) for v in (thk, vp, vs, rho)], drda, dadb

def exact(f):
    c = np.zeros(nt); ns = ctypes.c_long(0)
    flag = H.hs_swd_rootsearch(n, *[F(v) for v in f], nt, P(t), P(c), ctypes.byref(ns))
    return flag, c, ns.value

def kernels(f, c, drda, dadb):
    G = np.zeros((nt, 2 * n))
    for k in range(nt):
        ka, kb, kh, kr = (np.zeros(n) for _ in range(4))
        H.hs_sregn96(n, *[F(v) for v in f], ctypes.c_double(t[k]), ctypes.c_double(c[k]), P(ka), P(kb), P(kh), P(kr))
        G[k, :n] = kb + ka * dadb + kr * drda * dadb
        G[k, n:] = kh
    return G

tot = dict(items=0, same=0, declined=0, nev=0, maxrel=0.0, nev_exact=0)
hist = {}
for ch in range(nchain):
    x = xs[ch].copy()
    p = 0.5 * rng.standard_normal(2 * n)
    f, drda, dadb = f32model(x)
    flag, c, _ = exact(f)
    assert flag
    for s in range(nsteps):
        G = kernels(f, c, drda, dadb)
        xn = x + dt * p
        dx = xn - x
        fn, drda, dadb = f32model(xn)
        flag_e, ce, ns = exact(fn)
        dc = np.ascontiguousarray(G @ dx); l1 = np.ascontiguousarray(np.abs(G) @ np.abs(dx))
        cw = np.zeros(nt); nev = np.zeros(nt, dtype=np.int32); st = np.zeros(nt, dtype=np.int32)
        H.hs_warm_roots(n, *[F(v) for v in fn], nt, P(t), P(c), P(dc), P(l1), 0, 0, P(cw), I(nev), I(st))
        ok = st == 1
        tot["items"] += nt; tot["declined"] += int((~ok).sum()); tot["nev"] += int(nev.sum()); tot["nev_exact"] += ns
        if flag_e:
            rel = np.abs(cw[ok] - ce[ok]) / ce[ok]
            tot["same"] += int((cw[ok] == ce[ok]).sum())
            if rel.size: tot["maxrel"] = max(tot["maxrel"], float(rel.max()))
            if rel.size and rel.max() > 1.2e-6:
                print("chain", ch, "step", s, "rel", rel.max(), "pred err", np.abs(c + dc - ce).max())
        for v in nev: hist[int(v)] = hist.get(int(v), 0) + 1
        # continue the trajectory from the warm roots where accepted (as the device would), exact ones otherwise
        c = np.where(ok, cw, ce); x = xn; f = fn
print(tot, "evals/item warm %.2f exact %.2f" % (tot["nev"] / tot["items"], tot["nev_exact"] / tot["items"]))
print("nev histogram", dict(sorted(hist.items())))

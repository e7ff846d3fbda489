"""CPU simulation of the period-parallel reference-root search (ExactGroup, tests/hostsim build of the device math)
along leapfrog-like trajectories of the bench's models: warm-started roots (WarmSearch) -> groups of periods with run-up
-> compared bit for bit with the sequential reference-semantics search of the same model.
    python scripts/exact_sim.py [nchain] [nsteps] [dt] [G] [runup]
"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from oracle import oracle as O

H = ctypes.CDLL(os.path.join(ROOT, "tests", "hostsim", "libhostsim_swd.so"))
H.hs_sregn96.restype = ctypes.c_double
DP = ctypes.POINTER(ctypes.c_double); FP = ctypes.POINTER(ctypes.c_float); IP = ctypes.POINTER(ctypes.c_int)
P = lambda a: a.ctypes.data_as(DP); F = lambda a: a.ctypes.data_as(FP); I = lambda a: a.ctypes.data_as(IP)

nchain = int(sys.argv[1]) if len(sys.argv) > 1 else 16
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dt = float(sys.argv[3]) if len(sys.argv) > 3 else 0.02
G = int(sys.argv[4]) if len(sys.argv) > 4 else 5
RU = int(sys.argv[5]) if len(sys.argv) > 5 else 2
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
    flag = H.hs_swd_rootsearch_split(n, *[F(v) for v in f], nt, P(t), P(c), ctypes.byref(ns))
    return flag, c, ns.value


def kernels(f, c, drda, dadb):
    Gm = np.zeros((nt, 2 * n))
    for k in range(nt):
        ka, kb, kh, kr = (np.zeros(n) for _ in range(4))
        H.hs_sregn96(n, *[F(v) for v in f], ctypes.c_double(t[k]), ctypes.c_double(c[k]), P(ka), P(kb), P(kh), P(kr))
        Gm[k, :n] = kb + ka * dadb + kr * drda * dadb
        Gm[k, n:] = kh
    return Gm


FAST = os.environ.get("EXACT_SIM_FAST", "1") != "0"      # predictor = the true change of the roots + 5 % error instead of the kernels (40 eigenfunction passes per step)
bounds = bench.bounds_of(bench.true_model(n))


def mirror(x, p):
    for _ in range(64):
        over, under = x > bounds[:, 1], x < bounds[:, 0]
        if not (over.any() or under.any()):
            break
        x = np.where(over, 2 * bounds[:, 1] - x, x); x = np.where(under, 2 * bounds[:, 0] - x, x)
        p = np.where(over | under, -p, p)
    return x, p


tot = dict(items=0, same=0, declined_items=0, nev_warm=0, nev_exact=0, nev_full=0, maxrel=0.0, warm_declined=0)
causes = {}
for ch in range(nchain):
    x = np.clip(xs[ch], bounds[:, 0], bounds[:, 1])
    p = 0.5 * rng.standard_normal(2 * n)
    f, drda, dadb = f32model(x)
    flag, c, _ = exact(f)
    assert flag
    for s in range(nsteps):
        xn, p = mirror(x + dt * p, p)
        dx = xn - x
        fn, drda_n, dadb_n = f32model(xn)
        flag_e, ce, ns = exact(fn)
        if FAST and flag_e:
            dc = np.ascontiguousarray((ce - c) * (1.0 + 0.05 * rng.standard_normal(nt))); l1 = np.ascontiguousarray(2.0 * np.abs(dc) + 1e-4)
        else:
            Gm = kernels(f, c, drda, dadb)
            dc = np.ascontiguousarray(Gm @ dx); l1 = np.ascontiguousarray(np.abs(Gm) @ np.abs(dx))
        drda, dadb = drda_n, dadb_n
        cw = np.zeros(nt); nev = np.zeros(nt, dtype=np.int32); st = np.zeros(nt, dtype=np.int32)
        H.hs_warm_roots(n, *[F(v) for v in fn], nt, P(t), P(c), P(dc), P(l1), 0, 0, P(cw), I(nev), I(st))
        tot["nev_warm"] += int(nev.sum()); tot["nev_full"] += ns
        if not (st == 1).all() or not flag_e:
            tot["warm_declined"] += 1
            c = ce; x = xn; f = fn
            continue
        cx = np.zeros(nt); sx = np.zeros(nt, dtype=np.int32); ng = (nt + G - 1) // G
        nevx = np.zeros(ng, dtype=np.int32); cz = np.zeros(ng, dtype=np.int32)
        H.hs_exact_roots(n, *[F(v) for v in fn], nt, P(t), P(cw), 0, 0, G, RU, P(cx), I(sx), I(nevx), I(cz))
        tot["nev_exact"] += int(nevx.sum())
        ok = sx == 1
        tot["items"] += nt; tot["declined_items"] += int((~ok).sum())
        for v in cz[cz > 0]:
            causes[int(v)] = causes.get(int(v), 0) + 1
        tot["same"] += int((cx[ok] == ce[ok]).sum())
        if ok.any():
            rel = np.abs(cx[ok] - ce[ok]) / ce[ok]
            tot["maxrel"] = max(tot["maxrel"], float(rel.max()))
        # the device continues from the exact roots (croot), the full search's where declined
        c = np.where(ok, cx, ce); x = xn; f = fn
print(tot)
it = max(tot["items"], 1)
print("identical %.5f%%  declined items %.4f%%  evals/item: warm+check %.2f, exact stage %.2f, full search %.2f" % (
    100.0 * tot["same"] / max(it - tot["declined_items"], 1), 100.0 * tot["declined_items"] / it,
    tot["nev_warm"] / it, tot["nev_exact"] / it, tot["nev_full"] / it))
print("decline causes", causes)

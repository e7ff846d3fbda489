"""GPU: the period-parallel reference-root search (option swd_warm_exact) against the history-free full search on the
same trajectories: roots bit for bit, misfit / gradient, evaluations per item, time per evaluation.
    python scripts/exact_gpu.py [nchain] [nsteps] [dt] [G] [runup] [n]
"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from test_gpu_warm import _bench_joint, _leapfrog_move

nchain = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dt = float(sys.argv[3]) if len(sys.argv) > 3 else 0.02
G = int(sys.argv[4]) if len(sys.argv) > 4 else 5
RU = int(sys.argv[5]) if len(sys.argv) > 5 else 2
n = int(sys.argv[6]) if len(sys.argv) > 6 else 30
nt = 512
dev = torch.device("cuda")
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
bounds = bench.bounds_of(bench.true_model(n))
lo, hi = tt(bounds[:, 0]), tt(bounds[:, 1])

ref, t = _bench_joint(0, n=n)
ex, _ = _bench_joint(2, n=n)
ap, _ = _bench_joint(2, n=n)
cx = ex._ensure(n); ca = ap._ensure(n)
cx.set_option("swd_warm_exact", 1); cx.set_option("swd_exact_group", G); cx.set_option("swd_exact_runup", RU)
ca.set_option("swd_warm_exact", 0)
if os.environ.get("EXACT_TOL_E9"):
    cx.set_option("swd_exact_origin_tol_e9", int(os.environ["EXACT_TOL_E9"]))
x = tt(np.clip(bench.make_models(nchain, 991206, n), bounds[:, 0], bounds[:, 1]))
p = tt(0.5 * np.random.default_rng(7).standard_normal((nchain, 2 * n)))
tot = dict(roots=0, ident=0, ident_approx=0)
worst = dict(c=0.0, m=0.0, g=0.0, ca=0.0, ma=0.0, ga=0.0)
gq = []; gqa = []
for s in range(nsteps + 1):
    m0, g0, d0, f0 = ref.misfit_and_grad_device(x)
    m1, g1, d1, f1 = ex.misfit_and_grad_device(x)
    m2, g2, d2, f2 = ap.misfit_and_grad_device(x)
    torch.cuda.synchronize()
    ok = (f0 != 0)
    assert torch.equal(f0, f1) and torch.equal(f0, f2), s
    c0, c1, c2 = d0[ok][:, nt:], d1[ok][:, nt:], d2[ok][:, nt:]
    tot["roots"] += c0.numel(); tot["ident"] += int((c0 == c1).sum()); tot["ident_approx"] += int((c0 == c2).sum())
    relg = lambda a, b: ((a - b).abs().amax(dim=1) / b.abs().amax(dim=1))
    worst["c"] = max(worst["c"], float(((c1 - c0).abs() / c0).max())); worst["ca"] = max(worst["ca"], float(((c2 - c0).abs() / c0).max()))
    worst["m"] = max(worst["m"], float(((m1[ok] - m0[ok]).abs() / m0[ok]).max())); worst["ma"] = max(worst["ma"], float(((m2[ok] - m0[ok]).abs() / m0[ok]).max()))
    r1 = relg(g1[ok], g0[ok]); r2 = relg(g2[ok], g0[ok])
    worst["g"] = max(worst["g"], float(r1.max())); worst["ga"] = max(worst["ga"], float(r2.max()))
    gq.append(r1.cpu().numpy()); gqa.append(r2.cpu().numpy())
    # (large steps: the random momentum alone sets the step length -- a unit-mass kick with the real gradient leaves the bounds)
    x, p = _leapfrog_move(x, p, g0 if dt <= 0.01 else torch.zeros_like(g0), dt, lo, hi)
print(f"dt {dt} G {G} runup {RU}: {tot['roots']} roots; bit-identical exact mode {tot['ident']} ({100.0 * tot['ident'] / tot['roots']:.5f} %), "
      f"approximate mode {tot['ident_approx']} ({100.0 * tot['ident_approx'] / tot['roots']:.3f} %)")
gq = np.concatenate(gq); gqa = np.concatenate(gqa)
print("exact mode vs full search: roots %.2e  misfit %.2e  gradient max %.2e p99.9 %.2e" % (worst["c"], worst["m"], worst["g"], np.quantile(gq, 0.999)))
print("approx mode vs full search: roots %.2e  misfit %.2e  gradient max %.2e p99 %.2e p99.9 %.2e" % (worst["ca"], worst["ma"], worst["ga"], np.quantile(gqa, 0.99), np.quantile(gqa, 0.999)))
for name, c in (("exact", cx), ("approx", ca)):
    it = max(c.stat("swd_warm_items"), 1)
    print(f"{name}: warm evals/item {c.stat('swd_warm_secular_evals') / it:.2f}, exact-stage evals/item {c.stat('swd_exact_secular_evals') / it:.2f}, "
          f"handed back {c.stat('swd_warm_declined_chains')} (exact stage {c.stat('swd_exact_declined_chains')})")

# time per evaluation (consecutive small steps, no full search in between)
for name, j in (("full search", ref), ("exact", ex), ("approx", ap)):
    xx = x.clone(); pp = p.clone()
    m, g, d, f = j.misfit_and_grad_device(xx)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for s in range(10):
            xx, pp = _leapfrog_move(xx, pp, g if dt <= 0.01 else torch.zeros_like(g), dt, lo, hi)
            m, g, d, f = j.misfit_and_grad_device(xx)
        torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 10
    print(f"{name}: {el * 1e3:.3f} ms per evaluation of {nchain} chains")

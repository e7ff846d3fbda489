"""How far does THE REFERENCE differ from itself?  The reference's own src/SWD, compiled three ways by oracle/Makefile
(`make -C oracle ref ref_variants`: flang -O2 -- the oracle's pin --, -O0, and -O3 -march=native = the reference's own
Release flags with fused multiply-adds), evaluates the SWD half of Joint_RF_SWD.misfit_and_grad
(model/model_surf.py:155-228) on the same unsorted 30-layer models; the RF half (identical for all three) comes from the
oracle.  Prints, per pair of builds, how many chains have a root that differs and the joint gradient's relative difference.
Build container only (needs /root/reference).   python3 scripts/ref_selfdiff.py [nmodels] [seed] [noise] | file.npz (key x)"""
import multiprocessing as mp
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def variant_eval(args):
    """(kind, xs, t, rfpar, drf, dswd) -> [(misfit, grad, dsyn, flag)] with the reference's libsurf of that build."""
    kind, xs, t, rfpar, drf, dswd = args
    from oracle import oracle as O
    lib = O.ref_libsurf_variant(kind)
    j = O.Joint_RF_SWD(1.0, 1.0, O.ReceiverFunc(*rfpar), O.SurfWD(tRc=t, lib=lib))
    j.set_obsdata(drf, dswd)
    return [j.misfit_and_grad(x) for x in xs]


def evaluate(xs, t, rfpar, drf, dswd, kinds=("O2", "O0", "native"), nproc=None):
    """{kind: results}; every (kind, slice of the models) in a process of its own (one libsurf build per process)."""
    nproc = nproc or max(1, len(os.sched_getaffinity(0)))
    per = max(1, nproc // len(kinds))
    jobs = [(k, xs[q], t, rfpar, drf, dswd) for k in kinds for q in np.array_split(np.arange(len(xs)), per) if len(q)]
    with mp.get_context("spawn").Pool(nproc, maxtasksperchild=1) as pool:
        out = pool.map(variant_eval, jobs, chunksize=1)
    res = {k: [] for k in kinds}
    for (k, *_), r in zip(jobs, out):
        res[k] += r
    return res


def compare(res, nt, a, b):
    nroot = ndiff = 0; g = []; m = []
    for ra, rb in zip(res[a], res[b]):
        if not (ra[3] and rb[3]) or not (np.isfinite(ra[1]).all() and np.isfinite(rb[1]).all()):
            continue
        nroot += 1; ndiff += int((ra[2][nt:] != rb[2][nt:]).any())
        g.append(np.abs(ra[1] - rb[1]).max() / np.abs(rb[1]).max()); m.append(abs(ra[0] - rb[0]) / abs(rb[0]))
    return nroot, ndiff, np.array(g), np.array(m)


def main():
    import bench
    from oracle import oracle as O
    O.build(ref=False)
    n, nt = 30, 512
    t = np.linspace(5, 44, bench.NPER)
    rfpar = (bench.RAY_P, nt, 0.1, bench.GAUSS, bench.TSHIFT, bench.WATER, "P", "freq")
    x_true = bench.true_model(n)
    bounds = bench.bounds_of(x_true)
    jt = O.Joint_RF_SWD(1.0, 1.0, O.ReceiverFunc(*rfpar), O.SurfWD(tRc=t))
    drf, dswd, flag = jt.forward(x_true)
    if len(sys.argv) > 1 and sys.argv[1].endswith(".npz"):
        xs = np.load(sys.argv[1])["x"]
    else:
        nm = int(sys.argv[1]) if len(sys.argv) > 1 else 64
        seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
        noise = float(sys.argv[3]) if len(sys.argv) > 3 else 0.08
        rng = np.random.default_rng(seed)
        xs = np.clip(x_true[None, :] * (1 + noise * rng.standard_normal((nm, 2 * n))), bounds[:, 0], bounds[:, 1])
    res = evaluate(xs, t, rfpar, drf, dswd)
    for a, b in (("O0", "O2"), ("native", "O2")):
        nroot, ndiff, g, m = compare(res, nt, a, b)
        print(f"{a} vs {b}: {nroot} chains, {ndiff} with a differing root; gradient rel. diff max {g.max():.3g}, "
              f"above 1e-5: {(g > 1e-5).sum()}, above 1e-6: {(g > 1e-6).sum()}; misfit max {m.max():.3g}")


if __name__ == "__main__":
    main()

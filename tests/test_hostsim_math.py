"""CPU build of the device math headers (tests/hostsim/*.cpp, a TEST HARNESS -- never shipped and never a
fallback) against the oracle and the golden vectors: lets the lane-level algorithms of the HIP kernels be
checked in the GPU-less container."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostsim")


@pytest.fixture(scope="module")
def hs():
    libs = {}
    for name in ("rf", "swd"):
        so = os.path.join(HERE, f"libhostsim_{name}.so")
        # (RFS_HOSTSIM_CXXFLAGS="-fsanitize=undefined -fno-sanitize-recover=undefined": the device math under the CPU's sanitizer --
        # the GPU pool has none)
        subprocess.run(["g++", "-O2", "-fPIC", "-shared", "-std=c++17"] + os.environ.get("RFS_HOSTSIM_CXXFLAGS", "").split() +
                       ["-o", so, os.path.join(HERE, f"hostsim_{name}.cpp")], check=True)
        libs[name] = ctypes.CDLL(so)
    libs["swd"].hs_sregn96.restype = ctypes.c_double
    return libs


def rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300))


DP = ctypes.POINTER(ctypes.c_double)
FP = ctypes.POINTER(ctypes.c_float)
P = lambda a: a.ctypes.data_as(DP)
F = lambda a: a.ctypes.data_as(FP)


def test_rf_row_column_sweeps_equal_reference_partials(hs, orc, golden):
    """O(n) row/column sweeps (rf_math.hpp) == the reference's O(n^2) partial products, per frequency."""
    g = golden["rf_core_reference"]
    c = ctypes.c_double
    for name in ("yaml7_nt125", "grad30_nt512", "lvz30_0_nt512", "grad50_nt512"):
        thk, vs = np.ascontiguousarray(g[f"{name}/thk"]), np.ascontiguousarray(g[f"{name}/vs"])
        vp, rho, _, _ = orc.empirical_relation(vs)
        vp, rho = np.ascontiguousarray(vp), np.ascontiguousarray(rho)
        n = len(vs); q = np.full(n, 9999.)
        w, sigma = g[f"{name}/w"], float(g[f"{name}/sigma"])
        for i in range(0, len(w), 3):
            R21 = np.zeros(1, complex); R22 = np.zeros(1, complex)
            R21m = np.zeros((4, n), complex); R22m = np.zeros((4, n), complex)
            hs["rf"].hs_rf_response_par_all(n, P(thk), P(rho), P(vp), P(vs), P(q), P(q), c(float(g["ray_p"])),
                                            c(w[i]), c(-sigma), 1, P(R21), P(R22), P(R21m), P(R22m))
            assert rel(R21, g[f"{name}/R21"][i]) < 1e-10 and rel(R22, g[f"{name}/R22"][i]) < 1e-10
            assert rel(R21m, g[f"{name}/R21_m"][i]) < 1e-9 and rel(R22m, g[f"{name}/R22_m"][i]) < 1e-9


def test_rf_row_peeling_and_commutator_density_partial(hs):
    """Pass B's row peeling (option rf_row_peeling): the rows rebuilt from the final row with the inverse layer
    matrices equal the stored rows of the bottom-up sweep, and the density partial from the commutator of the layer
    matrix equals the closed form -- 30 and 50 layers, P and S type, teleseismic slownesses, the bench's frequency band."""
    import ctypes
    H = hs["rf"] if isinstance(hs, dict) else hs.rf
    H.hs_rf_peeling_errors.restype = None
    dp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    worst = [0.0, 0.0]
    for n, h in ((30, 2.0), (50, 1.2)):
        vs = np.linspace(2.0, 4.6, n); thk = np.full(n, h)
        vp = 0.9409 + 2.0947 * vs - 0.8206 * vs ** 2 + 0.2683 * vs ** 3 - 0.0251 * vs ** 4
        rho = 1.6612 * vp - 0.4721 * vp ** 2 + 0.0671 * vp ** 3 - 0.0043 * vp ** 4 + 0.000106 * vp ** 5
        q = np.full(n, 9999.0)
        for p in (0.045, 0.08):
            for rf_type in (1, 2):
                for kbin in (1, 17, 64, 127):
                    out = np.zeros(2)
                    H.hs_rf_peeling_errors(n, dp(thk), dp(rho), dp(vp), dp(vs), dp(q), dp(q), ctypes.c_double(p),
                                           ctypes.c_double(2 * np.pi * kbin / 51.2), ctypes.c_double(-4.0 / 51.2), rf_type, dp(out))
                    worst = [max(worst[0], out[0]), max(worst[1], out[1])]
    assert worst[0] < 1e-12 and worst[1] < 1e-11, worst


def test_rf_float32_step_stays_inside_its_margin(hs):
    """Pass A's float32 sweep of the frequencies beyond the gradient's band (rf_row_step_f32; host build: libm instead of
    the hardware exp2 / sin / cos): wherever the kernels would allow it -- growth exponent at most rf_f32_emax(n) -- the
    error of |R21|^2 stays far below RF_F32_MARGIN = 1e-2, which is what k_rf_mid1 adds to the float32 maximum before it
    lets it decide anything.  30 and 50 layers, ordered and random stacks, P and S type, slownesses up to 0.08 s/km,
    frequencies from the band limit to the Nyquist bins of the bench shapes."""
    import ctypes
    H = hs["rf"] if isinstance(hs, dict) else hs.rf
    H.hs_rf_f32_error.restype = None
    dp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    rng = np.random.default_rng(11)
    worst = 0.0; allowed = refused = 0
    for n, h in ((30, 2.0), (50, 1.2), (12, 5.0)):
        for trial in range(6):
            vs = np.linspace(2.0, 4.6, n) if trial == 0 else np.sort(1.8 + 3.0 * rng.random(n))
            if trial >= 4:
                vs = rng.permutation(vs)
            thk = np.full(n, h) * (1.0 if trial == 0 else 0.6 + 0.8 * rng.random(n))
            vp = 0.9409 + 2.0947 * vs - 0.8206 * vs ** 2 + 0.2683 * vs ** 3 - 0.0251 * vs ** 4
            rho = 1.6612 * vp - 0.4721 * vp ** 2 + 0.0671 * vp ** 3 - 0.0043 * vp ** 4 + 0.000106 * vp ** 5
            q = np.full(n, 9999.0)
            for p in (0.045, 0.065, 0.08):
                for rf_type in (1, 2):
                    for w in (15.7, 31.4, 62.8, 125.6):
                        for sigma in (4.0 / 51.2, 4.0 / 12.8):
                            out = np.zeros(3)
                            H.hs_rf_f32_error(n, dp(thk), dp(rho), dp(vp), dp(vs), dp(q), dp(q), ctypes.c_double(p),
                                              ctypes.c_double(w), ctypes.c_double(-sigma), rf_type, dp(out))
                            if out[1] <= out[2]:
                                allowed += 1; worst = max(worst, out[0])
                            else:
                                refused += 1
    assert allowed > 400 and refused > 0 and worst < 1e-3, (allowed, refused, worst)


def test_rf_float32_verdict_is_a_proof(hs):
    """rf_f32_decide (k_rf_mid1): random |R21|^2 spectra whose values beyond the band are only known to within the margin.
    Verdict 0 must mean that the band maxima ARE the maxima over all frequencies; verdict 1 that no band frequency is
    touched by the water level built from the TRUE maxima (neither fai nor the adjoint's fai2); everything else is swept
    again -- and all three verdicts occur."""
    import ctypes
    H = hs["rf"] if isinstance(hs, dict) else hs.rf
    H.hs_rf_f32_decide.restype = ctypes.c_int
    H.hs_rf_f32_decide.argtypes = [ctypes.c_double] * 7
    H.hs_rf_f32_margin.restype = ctypes.c_double
    margin = H.hs_rf_f32_margin()
    rng = np.random.default_rng(3)
    seen = {0: 0, 1: 0, 2: 0}
    for trial in range(4000):
        nk, n2 = 128, int(rng.choice([257, 1025]))
        spread = float(rng.choice([0.3, 1.0, 3.0, 8.0]))
        wa = np.exp(spread * rng.standard_normal(n2)) * (1.0 + 0.5 * np.cos(np.arange(n2) * rng.uniform(0.01, 0.3)))
        if rng.random() < 0.5:
            wa[0] *= 1.0 + 4.0 * rng.random()                  # (the bench shapes: maximum at DC)
        water = float(rng.choice([1e-4, 1e-3, 1e-2, 0.1, 0.5]))
        wb = wa * wa
        # what the float32 sweep may return: anything whose (1 + margin) multiple still covers the true value
        ha = wa[nk:] * (1.0 + (2.0 * rng.random(n2 - nk) - 1.0) * margin / (1.0 + margin) * 0.999)
        hb = ha * ha
        v = H.hs_rf_f32_decide(water, wa[:nk].max(), wb[:nk].max(), ha.max(), hb.max(), wa[:nk].min(), wb[:nk].min())
        seen[v] += 1
        if v == 0:
            assert wa.max() == wa[:nk].max() and wb.max() == wb[:nk].max()
        elif v == 1:
            assert np.all(wa[:nk] >= water * wa.max()) and np.all(wb[:nk] >= water * wb.max())
    assert min(seen.values()) > 50, seen


@pytest.mark.parametrize("entry", ["hs_swd_rootsearch", "hs_swd_rootsearch_split"])
def test_root_search_state_machine(hs, orc, golden, entry):
    """Request/advance state machine (+ the split secular function of the multi-lane kernels)."""
    g = golden["swd_reference"]
    nexact = ntot = 0
    for name in sorted({k.split("/")[0] for k in g.files}):
        thk, vs, t = g[f"{name}/thk"], g[f"{name}/vs"], np.ascontiguousarray(g[f"{name}/t"])
        vp, rho, _, _ = orc.empirical_relation(vs)
        f = [np.ascontiguousarray(x.astype(np.float32)) for x in (thk, vp, vs, rho)]
        cg = np.zeros(len(t)); ns = ctypes.c_long(0)
        flag = getattr(hs["swd"], entry)(len(vs), *[F(x) for x in f], len(t), P(t), P(cg), ctypes.byref(ns))
        ref = g[f"{name}/Rc/fwd_c"]
        assert bool(flag) == bool(g[f"{name}/Rc/fwd_flag"]), name
        if entry == "hs_swd_rootsearch":
            assert np.array_equal(cg, ref), name              # same arithmetic as the reference: bit-exact
        else:                                                 # reciprocal-multiply form: reference's own 1e-6 c tolerance
            assert np.all(np.abs(cg - ref) <= 1.2e-6 * np.abs(ref)), name
        nexact += int((cg == ref).sum()); ntot += len(ref)
    assert nexact >= 0.99 * ntot


@pytest.mark.parametrize("entry", ["hs_swd_rootsearch", "hs_swd_rootsearch_split"])
def test_a_model_that_is_no_model_fails_at_the_first_evaluation(hs, orc, entry):
    """RootSearchT::begin (round 6): a model whose fastest layer lies more than SWD_MAX_SCAN above the start value -- a position
    that left its bounds -- would have the reference's scan walk millions of cells of 0.005 km/s
    (half a minute on the device, with every other chain of the batch waiting).  The search fails at its first evaluation: flag
    0, zeros; an ordinary model beside it is searched as ever."""
    thk = np.array([3., 3, 4, 5, 5, 6, 7, 8, 10, 0]); vs = np.linspace(2.9, 4.6, 10)
    t = np.ascontiguousarray(np.arange(5., 41.))
    for scale, nanat in ((2.0e4, None), (1.0, 3)):
        v = vs * scale
        if nanat is not None:
            v = v.copy(); v[nanat] = np.nan
        vp, rho, _, _ = orc.empirical_relation(np.where(np.isfinite(v), np.minimum(v, 8.0), 4.0))
        f = [np.ascontiguousarray(x.astype(np.float32)) for x in (thk, vp, v, rho)]
        cg = np.full(len(t), -1.0); ns = ctypes.c_long(0)
        flag = getattr(hs["swd"], entry)(len(v), *[F(x) for x in f], len(t), P(t), P(cg), ctypes.byref(ns))
        # (a not-a-number among the velocities is not seen by the extremal-velocity comparisons: that search runs, and fails in
        # bounded time of its own accord)
        assert flag == 0 and not cg.any() and (ns.value == 1 if nanat is None else ns.value < 20000), (scale, nanat, flag, ns.value)
    vp, rho, _, _ = orc.empirical_relation(vs)
    f = [np.ascontiguousarray(x.astype(np.float32)) for x in (thk, vp, vs, rho)]
    cg = np.zeros(len(t)); ns = ctypes.c_long(0)
    assert getattr(hs["swd"], entry)(len(vs), *[F(x) for x in f], len(t), P(t), P(cg), ctypes.byref(ns)) == 1 and (cg > 2.0).all()


def test_fused_eigenfunction_sweep(hs, orc, golden):
    g = golden["swd_reference"]
    c = ctypes.c_double
    for name in ("yaml7", "grad30", "lvz30_1", "prior30_2", "grad50"):
        thk, vs, t = g[f"{name}/thk"], g[f"{name}/vs"], g[f"{name}/t"]
        vp, rho, _, _ = orc.empirical_relation(vs)
        f = [np.ascontiguousarray(x.astype(np.float32)) for x in (thk, vp, vs, rho)]
        n = len(vs)
        for i in range(0, len(t), 5):
            k = [np.zeros(n) for _ in range(4)]
            hs["swd"].hs_sregn96(n, *[F(x) for x in f], c(t[i]), c(g[f"{name}/Rc/c"][i]), *[P(x) for x in k])
            for got, key in zip(k, ("dcda", "dcdb", "dcdh", "dcdr")):
                assert rel(got, g[f"{name}/Rc/{key}"][i]) < 1e-8, (name, i, key)


def test_love_and_sphere_device_math(hs, golden):
    """Host build of the Love / earth-flattening device math (swd_math.hpp: swd_secular_love, swd_flatten_f32,
    swd_bldsph, sl_up, sl_down_energy, sr_tm) against the compiled reference's fixtures: flat roots from the
    request/advance state machine, kernels at the reference's own roots."""
    g = golden["swd_love_sphere_reference"]
    from oracle import oracle as O
    H = hs["swd"]
    H.hs_eigen_general.restype = ctypes.c_double
    nroot = nexact = 0
    for name in sorted({k.split("/")[0] for k in g.files if k.endswith("/thk")}):
        thk, vs, t = g[f"{name}/thk"], g[f"{name}/vs"], np.ascontiguousarray(g[f"{name}/t"])
        vp, rho, _, _ = O.empirical_relation(vs)
        h, a, b, r = [np.ascontiguousarray(np.asarray(x, dtype=np.float64).astype(np.float32)) for x in (thk, vp, vs, rho)]
        n, nt = len(h), len(t)
        for love, wt in ((0, "Rc"), (1, "Lc")):
            for sph in (0, 1):
                if not love and not sph:
                    continue
                key = f"{name}/{wt}/{sph}"
                c = np.zeros(nt)
                flag = H.hs_rootsearch_general(n, F(h), F(a), F(b), F(r), nt, P(t), P(c), love, sph)
                assert bool(flag) == bool(g[f"{key}/flag"]), key
                if not flag:
                    continue
                # adjoint_kernel's c is the spherical phase velocity c_flat / tm: convert the flat roots through the
                # eigen entry below and compare there; count bit-exact matches of the converted value
                for k in range(0, nt, 5):
                    ka, kb, kh, kr = (np.zeros(n) for _ in range(4))
                    cp = ctypes.c_double(c[k])
                    H.hs_eigen_general(n, F(h), F(a), F(b), F(r), ctypes.c_double(t[k]), ctypes.byref(cp),
                                       P(ka), P(kb), P(kh), P(kr), love, sph)
                    nroot += 1
                    nexact += int(cp.value == g[f"{key}/c"][k])
                    assert abs(cp.value - g[f"{key}/c"][k]) <= 1.2e-6 * cp.value, key     # nevill's own tolerance
                    if cp.value != g[f"{key}/c"][k]:
                        continue                                                          # kernels at another root
                    for arr, kk in ((ka, "dcda"), (kb, "dcdb"), (kr, "dcdr"), (kh, "dcdh")):
                        if love and kk == "dcda":
                            continue
                        ref = g[f"{key}/{kk}"][k]
                        assert np.abs(arr - ref).max() <= 2e-8 * np.abs(ref).max(), (key, kk, k)
    assert nroot > 100 and nexact >= 0.97 * nroot


@pytest.mark.parametrize("nseg", [1, 2, 4])
def test_family_form_of_the_search_rayleigh_and_love(hs, golden, nseg):
    """The arithmetic of the lanes-per-item kernel (k_swd_roots_split) for both wave families on the host: per-layer
    entries through SwdRayFamily / SwdLoveFamily, raw recurrence with power-of-two rescales, sequential or in nseg
    segments started from unit vectors and folded.  Same flags as the compiled reference, flat roots within nevill's own
    tolerance, almost all of them the very same float32 value as the plain form finds."""
    g = golden["swd_love_sphere_reference"]
    from oracle import oracle as O
    H = hs["swd"]
    nroot = nsame = 0
    for name in sorted({k.split("/")[0] for k in g.files if k.endswith("/thk")}):
        thk, vs, t = g[f"{name}/thk"], g[f"{name}/vs"], np.ascontiguousarray(g[f"{name}/t"])
        vp, rho, _, _ = O.empirical_relation(vs)
        h, a, b, r = [np.ascontiguousarray(np.asarray(x, dtype=np.float64).astype(np.float32)) for x in (thk, vp, vs, rho)]
        n, nt = len(h), len(t)
        for love, wt in ((0, "Rc"), (1, "Lc")):
            for sph in (0, 1):
                key = f"{name}/{wt}/{sph}"
                if f"{key}/flag" not in g.files:
                    continue
                c0, c1 = np.zeros(nt), np.zeros(nt)
                f0 = H.hs_rootsearch_general(n, F(h), F(a), F(b), F(r), nt, P(t), P(c0), love, sph)
                f1 = H.hs_rootsearch_family(n, F(h), F(a), F(b), F(r), nt, P(t), P(c1), love, sph, nseg)
                assert bool(f1) == bool(f0) == bool(g[f"{key}/flag"]), key
                if not f1:
                    continue
                assert np.all(np.abs(c1 - c0) <= 1.2e-6 * np.abs(c0)), (key, nseg)
                nroot += nt; nsame += int((c1 == c0).sum())
    assert nroot > 500 and nsame >= 0.99 * nroot, (nroot, nsame)


@pytest.mark.parametrize("group,runup", [(5, 2), (8, 2), (3, 3), (40, 0)])
def test_exact_group_search_gives_the_reference_roots(hs, golden, group, runup):
    """ExactGroup (swd_math.hpp, the lane code of k_swd_exact): from roots that are only CONVERGED (the sign change, as the
    warm start leaves them: 0.5 .. 1e-6 c above what the reference's nevill returns) to the reference's own roots, periods
    in groups with run-up.  Rayleigh and Love, flat and flattened earth, the fixtures' models: the float32 results are
    those of the sequential reference-semantics search, bit for bit (group = 40, run-up 0: one lane walks the whole
    sequence -- the hand-over is then the reference's own and every root must be identical)."""
    g = golden["swd_love_sphere_reference"]
    from oracle import oracle as O
    H = hs["swd"]
    I = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
    rng = np.random.default_rng(group * 10 + runup)
    nroot = nsame = ndecl = 0
    for name in sorted({k.split("/")[0] for k in g.files if k.endswith("/thk")}):
        thk, vs, t = g[f"{name}/thk"], g[f"{name}/vs"], np.ascontiguousarray(g[f"{name}/t"])
        vp, rho, _, _ = O.empirical_relation(vs)
        h, a, b, r = [np.ascontiguousarray(np.asarray(x, dtype=np.float64).astype(np.float32)) for x in (thk, vp, vs, rho)]
        n, nt = len(h), len(t)
        for love, wt in ((0, "Rc"), (1, "Lc")):
            for sph in (0, 1):
                key = f"{name}/{wt}/{sph}"
                if f"{key}/flag" not in g.files:
                    continue
                c0 = np.zeros(nt)
                if not H.hs_rootsearch_family(n, F(h), F(a), F(b), F(r), nt, P(t), P(c0), love, sph, 1):
                    continue
                # converged roots: the reference's + 0.5 .. 1e-6 c, float32-rounded like k_swd_warm's output
                approx = np.ascontiguousarray((c0 * (1.0 + rng.uniform(5e-7, 1e-6, nt))).astype(np.float32).astype(np.float64))
                cx = np.zeros(nt); st = np.zeros(nt, dtype=np.int32)
                ng = (nt + group - 1) // group
                nev = np.zeros(ng, dtype=np.int32); cause = np.zeros(ng, dtype=np.int32)
                H.hs_exact_roots(n, F(h), F(a), F(b), F(r), nt, P(t), P(approx), love, sph, group, runup, P(cx), I(st), I(nev), I(cause))
                ok = st == 1
                ndecl += int((~ok).sum())
                assert np.all(np.abs(cx[ok] - c0[ok]) <= 1.2e-6 * c0[ok]), key
                nroot += int(ok.sum()); nsame += int((cx[ok] == c0[ok]).sum())
                if runup == 0 and group >= nt:
                    assert np.array_equal(cx[ok], c0[ok]), key
    # (declined groups: a root within one scan step of the start value or of the fastest layer -- the full search's business)
    assert nroot > 500 and ndecl <= 0.1 * nroot, (nroot, ndecl)
    assert nsame >= (0.995 if runup >= 2 else 0.98) * nroot, (nroot, nsame)


def test_lazy_nevill_takes_the_same_path_as_the_full_one(hs):
    """CellNevillT<true> supplies the values of far-side bisection points itself where the test they feed is won by a factor
    of three (swd_math.hpp); CellNevillT<false> evaluates every point like the reference.  Sorted, unsorted ("wild"),
    low-velocity-layer and thin-layer models, Rayleigh and Love, 40 periods in one sequential group: the same float32
    roots from both, at roughly half the evaluations."""
    import bench
    from oracle import oracle as O
    H = hs["swd"]
    I = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
    n, nt = 30, 40
    t = np.ascontiguousarray(np.linspace(5, 44, nt))
    rng = np.random.default_rng(11)
    xs = bench.make_models(192, 123, n)
    nroot = nev_l = nev_p = 0
    for i, x in enumerate(xs):
        vs, thk = x[:n].copy(), x[n:].copy()
        if i % 4 == 1: vs = rng.permutation(vs)
        if i % 4 == 2: vs[rng.integers(2, n - 2)] *= 0.8
        if i % 4 == 3: thk *= 0.5
        vp, rho, _, _ = O.empirical_relation(vs)
        h, a, b, r = [np.ascontiguousarray(np.asarray(v, dtype=np.float64).astype(np.float32)) for v in (thk, vp, vs, rho)]
        for love in (0, 1):
            c0 = np.zeros(nt)
            if not H.hs_rootsearch_family(n, F(h), F(a), F(b), F(r), nt, P(t), P(c0), love, 0, 1):
                continue
            approx = np.ascontiguousarray((c0 * (1.0 + rng.uniform(5e-7, 1e-6, nt))).astype(np.float32).astype(np.float64))
            res = []
            for lazy in (1, 0):
                cx = np.zeros(nt); st = np.zeros(nt, dtype=np.int32); nev = np.zeros(1, dtype=np.int32); cause = np.zeros(1, dtype=np.int32)
                ns = ctypes.c_long(0)
                H.hs_exact_roots2(n, F(h), F(a), F(b), F(r), nt, P(t), P(approx), love, 0, nt, 0, lazy, P(cx), I(st), I(nev), I(cause), ctypes.byref(ns))
                res.append((cx, st, int(nev[0])))
            (cl, sl, nl), (cp, sp, npl) = res
            assert np.array_equal(sl, sp) and np.array_equal(cl, cp), (i, love)
            if (sp == 1).all():
                nroot += nt; nev_l += nl; nev_p += npl
    assert nroot > 10000 and nev_l <= 0.62 * nev_p, (nroot, nev_l, nev_p)


@pytest.mark.parametrize("budget", [1, 3, 7])
def test_reference_root_machine_handed_from_lane_to_lane(hs, budget):
    """k_swd_exact in rounds (round 6): a lane that has used up its budget of evaluations saves its whole machine --
    ExactGroupT::save: 68 doubles incl. the Neville table, in the middle of a period if need be -- and k_swd_exact_coop loads and
    continues it.  The lane code on the host: groups of 4 periods behind 2 run-up periods, handed over every `budget` evaluations
    into a machine that starts from garbage, give the uninterrupted run's roots, verdicts and evaluation counts."""
    H = hs["swd"]
    I = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
    from oracle import oracle as O
    rng = np.random.default_rng(40 + budget)
    nroots = nmoves = 0
    for trial in range(10):
        n = int(rng.integers(4, 20))
        thk = 2.0 + 6.0 * rng.random(n); thk[-1] = 0.0
        vs = np.sort(2.6 + 1.9 * rng.random(n))
        if trial % 3 == 2:
            vs[1:-1] = rng.permutation(vs[1:-1])
        vp, rho, _, _ = O.empirical_relation(vs)
        h, a, b, r = [np.ascontiguousarray(np.asarray(x, dtype=np.float64).astype(np.float32)) for x in (thk, vp, vs, rho)]
        t = np.ascontiguousarray(np.sort(5.0 + 30.0 * rng.random(14)))
        nt = len(t)
        c0 = np.zeros(nt)
        if not H.hs_rootsearch_family(n, F(h), F(a), F(b), F(r), nt, P(t), P(c0), 0, 0, 1):
            continue
        approx = np.ascontiguousarray((c0 * (1.0 + rng.uniform(5e-7, 1e-6, nt))).astype(np.float32).astype(np.float64))
        ng = (nt + 3) // 4
        c1 = np.zeros(nt); s1 = np.zeros(nt, dtype=np.int32); n1 = np.zeros(ng, dtype=np.int32)
        c2 = np.zeros(nt); s2 = np.zeros(nt, dtype=np.int32); n2 = np.zeros(ng, dtype=np.int32)
        H.hs_exact_roots_handover(n, F(h), F(a), F(b), F(r), nt, P(t), P(approx), 4, 2, 1 << 30, P(c1), I(s1), I(n1))
        nmoves += H.hs_exact_roots_handover(n, F(h), F(a), F(b), F(r), nt, P(t), P(approx), 4, 2, budget, P(c2), I(s2), I(n2))
        assert np.array_equal(c1, c2) and np.array_equal(s1, s2) and np.array_equal(n1, n2), trial
        nroots += int(s1.sum())
    assert nroots > 60 and nmoves > 100, (nroots, nmoves)


@pytest.mark.parametrize("budget", [1, 2, 3])
def test_warm_search_handed_from_lane_to_lane(hs, budget):
    """k_swd_warm runs in rounds: a search that has used its budget of evaluations is written out (12 doubles + the word of
    WarmSearch::pack_small) and picked up by a lane of the next launch.  The lane code on the host: searches from good,
    poor and hopeless predictions (Newton start, bracket, widened bracket, bisection, failure), handed over every `budget`
    evaluations into a machine that starts from garbage, end where the uninterrupted search ends -- root, evaluation count
    and verdict."""
    H = hs["swd"]
    I = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
    from oracle import oracle as O
    rng = np.random.default_rng(11 + budget)
    nroots = nmoves = nlong = 0
    for trial in range(12):
        n = int(rng.integers(4, 16))
        thk = 2.0 + 6.0 * rng.random(n); thk[-1] = 0.0
        vs = np.sort(2.6 + 1.9 * rng.random(n))
        if trial % 3 == 2:
            vs[1:-1] = rng.permutation(vs[1:-1])                     # velocity inversions: crowded spectra
        vp, rho, _, _ = O.empirical_relation(vs)
        h, a, b, r = [np.ascontiguousarray(np.asarray(x, dtype=np.float64).astype(np.float32)) for x in (thk, vp, vs, rho)]
        t = np.ascontiguousarray(np.sort(5.0 + 30.0 * rng.random(12)))
        nt = len(t)
        c0 = np.zeros(nt)
        if not H.hs_rootsearch_family(n, F(h), F(a), F(b), F(r), nt, P(t), P(c0), 0, 0, 1):
            continue
        # predictions: the root moved by up to 0.1 %, 1 % or 5 %, with a first-order estimate that is right, off, or useless
        move = c0 * rng.choice([1e-3, 1e-2, 5e-2], nt) * rng.uniform(-1, 1, nt)
        cprev = np.ascontiguousarray(c0 - move)
        dc = np.ascontiguousarray(move * rng.choice([1.0, 0.7, -0.5], nt))
        l1 = np.ascontiguousarray(np.abs(move) * rng.uniform(1.0, 3.0, nt))
        c1 = np.zeros(nt); n1 = np.zeros(nt, dtype=np.int32); s1 = np.zeros(nt, dtype=np.int32)
        c2 = np.zeros(nt); n2 = np.zeros(nt, dtype=np.int32); s2 = np.zeros(nt, dtype=np.int32)
        H.hs_warm_roots_handover(n, F(h), F(a), F(b), F(r), nt, P(t), P(cprev), P(dc), P(l1), 1 << 30, P(c1), I(n1), I(s1))
        nmoves += H.hs_warm_roots_handover(n, F(h), F(a), F(b), F(r), nt, P(t), P(cprev), P(dc), P(l1), budget, P(c2), I(n2), I(s2))
        assert np.array_equal(c1, c2) and np.array_equal(n1, n2) and np.array_equal(s1, s2), trial
        nroots += int(s1.sum()); nlong += int((n1 > 8).sum())
    assert nroots > 60 and nmoves > 100 and nlong > 5, (nroots, nmoves, nlong)


def test_fast_exp_and_sincos_accuracy(tmp_path):
    """fm_exp / fm_sincos (cplx.hpp: the transcendental functions of every layer sweep) against long double libm
    on the ranges they are used on: < 1 ulp (exp) and < 1.5 ulp (sin, cos); absolute error at multiples of pi/2."""
    src = tmp_path / "fm.cpp"
    src.write_text(r'''
#include <cstdio>
#include <cmath>
#include <random>
#include "%s/rfsurfhmc_amd/csrc/cplx.hpp"
using namespace rfs;
int main() {
    std::mt19937_64 g(1);
    const double h = 1.1102230246251565e-16;
    double we = 0, ws = 0, wc = 0, wa = 0;
    for (int i = 0; i < 2000000; i++) {
        double u = (double)g() / 1.8446744073709552e19;
        double x = -120.0 + 170.0 * u;
        long double ref = expl((long double)x);
        we = fmax(we, fabs((double)((fm_exp(x) - ref) / ref)) / h);
        double scale = (i %% 3 == 0) ? 3.0 : (i %% 3 == 1 ? 300.0 : 1.0e5);
        double y = (2 * u - 1) * scale, s, c;
        fm_sincos(y, &s, &c);
        long double rs = sinl((long double)y), rc = cosl((long double)y);
        if (fabsl(rs) > 1e-3L) ws = fmax(ws, fabs((double)((s - rs) / rs)) / h);
        if (fabsl(rc) > 1e-3L) wc = fmax(wc, fabs((double)((c - rc) / rc)) / h);
        wa = fmax(wa, fmax(fabs((double)(s - rs)), fabs((double)(c - rc))));
    }
    printf("%%.4f %%.4f %%.4f %%.3e\n", we / 2, ws / 2, wc / 2, wa);
    return 0;
}
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    exe = tmp_path / "fm"
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", str(src), "-o", str(exe)], check=True)
    ue, us, uc, ab = map(float, subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split())
    assert ue < 1.0 and us < 1.5 and uc < 1.5 and ab < 2.5e-16, (ue, us, uc, ab)

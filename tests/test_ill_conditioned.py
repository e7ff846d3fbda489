"""The 1e-5 gradient tail of the measured mode, explained (VERDICT r05 "Next 1").

tests/golden/ill_conditioned_reference.npz (oracle/make_golden.py --ill-conditioned): 165 of the 659 burned-in chains -- out of
12 288 mid-trajectory chains of the bench's own sampler run, device steps 200 / 350 / 500 / 650 -- on which the device held a
root that is not the flang -O2 reference's, evaluated by THREE builds of the reference's own src/SWD under the reference's own
plugins: flang -O2 (the oracle's pin), -O0, and -O3 -march=native (the reference's Release flags, CMakeLists.txt:16-34: fused
multiply-adds), plus the gradient and the roots the device used.  What the file shows, asserted here on the CPU:

  1. the oracle (C restatement) IS the -O2 build on these chains too: roots bit for bit, gradient to 1e-9;
  2. the reference differs from ITSELF there by more than the contract: native against -O2 up to 1.9e-5 in the joint
     gradient, 11 chains above 1e-5, a differing root on 404 of the 659 chains;
  3. every root the device held lies within the reference's own refinement tolerance (1e-6 c, surfdisp96.f:627, plus the
     float32 rounding of :302) of the -O2 build's, and most of those that differ are the native build's roots bit for bit;
  4. the device's gradient equals the reference's eigenfunction pass evaluated AT THE DEVICE'S ROOTS to 1e-9 on every chain:
     all of its deviation from the -O2 build (up to 2.2e-5) is which end of that 1e-6 c bracket a root sits on.
"""
import os

import numpy as np
import pytest

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ill_conditioned_reference.npz")
RFPAR = (0.045, 512, 0.1, 1.5, 5.0, 0.001, "P", "freq")


def rel_rows(a, b):
    return np.abs(a - b).max(axis=1) / np.abs(b).max(axis=1)


@pytest.fixture(scope="module")
def fix():
    return np.load(FIX)


def test_the_reference_differs_from_itself_by_more_than_1e5_on_these_chains(fix):
    nat, o0, dev = fix["stats/native_vs_O2"], fix["stats/O0_vs_O2"], fix["stats/device_vs_O2"]
    # [chains evaluated, chains with a differing root, max gradient difference, chains above 1e-5, above 1e-6]
    assert nat[0] == 659 and nat[1] >= 400 and nat[2] > 1.5e-5 and nat[3] >= 10
    assert o0[1] >= 1 and o0[2] < 1e-5                       # even -O0 against -O2 moves roots (8 chains), by less
    assert dev[2] < 1.25 * nat[2] and dev[3] <= 2.5 * nat[3]    # the device: the same scatter (2.2e-5, 25 chains)
    # ... and on the chains the file keeps, recomputed from the stored gradients
    g2, gn = fix["O2/joint_grad_hybrid"], fix["native/joint_grad_hybrid"]
    sd = rel_rows(gn, g2)
    assert sd.max() == pytest.approx(nat[2], rel=1e-12) and int((sd > 1e-5).sum()) == int(nat[3])
    same_root = (fix["native/roots"] == fix["O2/roots"]).all(axis=1)
    assert sd[same_root].max() < 1e-6                        # with the same roots two builds agree: the roots are the cause
    assert rel_rows(fix["native/swd_grad"], fix["O2/swd_grad"])[same_root].max() < 1e-6


def test_device_roots_lie_inside_the_reference_own_bracket(fix):
    cd, c2, cn = fix["device_roots"], fix["O2/roots"], fix["native/roots"]
    assert (np.abs(cd - c2) / c2).max() <= 1.2e-6            # nevill's tolerance 1e-6 c + float32 rounding (measured 1.07e-6)
    assert (np.abs(cn - c2) / c2).max() <= 1.2e-6            # ... which is also how far two builds of the reference are apart
    dn = (cn != c2)
    assert dn.sum() >= 100 and ((cd == cn) & dn).sum() >= 0.7 * dn.sum()      # where native leaves -O2, the device mostly goes with it (103 of 132 roots)
    assert np.array_equal(cd, cd.astype(np.float32).astype(np.float64))       # float32 values (surfdisp96.f:302)


def test_oracle_is_the_O2_build_and_device_gradient_is_the_reference_at_its_roots(fix, orc):
    O = orc
    x, t, dobs, nt = fix["x"], fix["t"], fix["dobs"], int(fix["nt"])
    rf = O.ReceiverFunc(*RFPAR); rf.set_obsdata(dobs[:nt])
    sw = O.SurfWD(tRc=t); sw.set_obsdata(dobs[nt:])
    wt = nt / len(t)                                          # model_rf_swd_vs_thk.py:79 with sigma1 = sigma2
    worst_o2 = worst_dev = worst_joint = 0.0
    rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
    for i in range(0, len(x), 2):                             # (every other chain: ~8 s)
        ms, gs, ds, flag = sw.misfit_and_grad(x[i])
        assert flag and np.array_equal(ds, fix["O2/roots"][i])
        worst_o2 = max(worst_o2, rel(gs, fix["O2/swd_grad"][i]))
        mr, gr, dr = rf.misfit_and_grad(x[i])
        worst_joint = max(worst_joint, rel(gr + wt * gs, fix["O2/joint_grad_hybrid"][i]))
        m2, gsd = O.swd_misfit_and_grad_at_roots(x[i], t, fix["device_roots"][i], dobs[nt:])
        worst_dev = max(worst_dev, rel(gr + wt * gsd, fix["device_joint_grad"][i]))
    assert worst_o2 <= 1e-9 and worst_joint <= 1e-9, (worst_o2, worst_joint)       # measured 3e-13
    assert worst_dev <= 1e-9, worst_dev                                            # measured 5e-11

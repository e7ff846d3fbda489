"""Test helper: the reference's NaN kernels.  Where a float32 Rayleigh phase velocity EQUALS a layer's float32 S or P velocity,
sregn96 divides by a zero vertical wavenumber and every kernel of that period is NaN (tests/golden/exact_equality_reference.npz,
DESIGN section 2): the gradient of such a chain is NaN, its flag True.  About one chain in 8192 per evaluation of a 30-layer
batch -- full-size tests have to expect them, and check that NaN appears nowhere else."""
import numpy as np


def exact_equality_rows(x, roots, n):
    """x [nchain, 2n] models, roots [nchain, nper] Rayleigh phase velocities (float32 values) -> bool [nchain]: some root
    equals a layer's float32 vs, or lies within a float32 step of its float32 vp (the polynomial vp(vs) is re-evaluated here in
    numpy: the device's f64 value may differ in the last bit before the cast)."""
    x = np.asarray(x, dtype=np.float64); roots = np.asarray(roots, dtype=np.float64)
    vs = x[:, :n]
    vp = 0.9409 + 2.0947 * vs - 0.8206 * vs ** 2 + 0.2683 * vs ** 3 - 0.0251 * vs ** 4
    vs32 = vs.astype(np.float32).astype(np.float64)
    vp32 = vp.astype(np.float32).astype(np.float64)
    hit = np.zeros(len(x), dtype=bool)
    for k in range(roots.shape[1]):
        c = roots[:, k:k + 1]
        hit |= (c == vs32).any(axis=1) | (np.abs(c - vp32) <= 2.4e-7 * vp32).any(axis=1)
    return hit


def check_nan_gradients(x, grad, roots, n, what=""):
    """NaN gradient rows are exactly explained by an exact equality; returns the mask of NaN rows."""
    grad = np.asarray(grad)
    bad = ~np.isfinite(grad).all(axis=1)
    if bad.any():
        hit = exact_equality_rows(np.asarray(x)[bad], np.asarray(roots)[bad], n)
        assert hit.all(), (what, "NaN gradient without a root that equals a layer velocity", int((~hit).sum()))
        assert bad.sum() <= max(4, len(grad) // 1000), (what, int(bad.sum()))
    return bad


def same(a, b):
    """array_equal with NaN == NaN (numpy arrays or torch tensors)."""
    try:
        import torch
        if isinstance(a, torch.Tensor):
            return bool(torch.equal(torch.nan_to_num(a, nan=-1.25e300), torch.nan_to_num(b, nan=-1.25e300))) if a.is_floating_point() \
                else bool(torch.equal(a, b))
    except ImportError:
        pass
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and bool(np.all((a == b) | ((a != a) & (b != b))))

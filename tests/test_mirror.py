"""Mirror reflection at the bounds (pyhmc/hmc.py:121-137, hmcda.py:152-168): the host form of the rule the device's flow_mirror
follows -- 64 reflections as the reference makes them, then the closed form."""
import numpy as np


def _reference_loop(x, p, b):
    x, p = x.copy(), p.copy()
    hi, lo = b[:, 1], b[:, 0]
    i1, i2 = x > hi, x < lo
    while (i1 | i2).any():
        x[i1] = 2 * hi[i1] - x[i1]; p[i1] = -p[i1]
        x[i2] = 2 * lo[i2] - x[i2]; p[i2] = -p[i2]
        i1, i2 = x > hi, x < lo
    return x, p


def test_mirror_is_the_references_loop_and_ends_for_any_point():
    from rfsurfhmc_amd.pyhmc.hmcda import _mirror
    rng = np.random.default_rng(3)
    b = np.array([[2.0, 5.0], [1.0, 3.0], [0.5, 0.75], [0.0, 0.0]])
    # points a few reflections out: the reference's loop, bit for bit
    x = np.column_stack([b[:3, 0][None, :] + (b[:3, 1] - b[:3, 0])[None, :] * (8 * rng.random((200, 3)) - 4), np.zeros((200, 1))])
    p = rng.standard_normal(x.shape)
    xn, pn = _mirror(x, p, b)
    for i in range(len(x)):
        xr, pr = _reference_loop(x[i], p[i], b)
        assert np.array_equal(xn[i], xr) and np.array_equal(pn[i], pr), i
    # points thousands of reflections out (a momentum that has blown up): inside the bounds, the reference's point up to rounding
    xb = np.array([[1234.56, -777.1, 3000.1, 0.0], [6.4e4, 9.9e5, -4.0e6, 0.0]])
    pb = np.ones_like(xb)
    xn, pn = _mirror(xb, pb, b)
    assert np.all(xn[:, :3] >= b[:3, 0]) and np.all(xn[:, :3] <= b[:3, 1])
    xr, pr = _reference_loop(xb[0], pb[0], b)
    assert np.allclose(xn[0], xr, rtol=0, atol=1e-9) and np.array_equal(pn[0], pr)
    # ... and points no loop would ever bring back
    xi = np.array([[np.inf, -np.inf, np.nan, 0.0], [1e305, -1e308, 4.0, 0.0]])
    xn, pn = _mirror(xi, np.ones_like(xi), b)
    assert np.isfinite(xn).all() and np.all(xn[:, :3] >= b[:3, 0]) and np.all(xn[:, :3] <= b[:3, 1])

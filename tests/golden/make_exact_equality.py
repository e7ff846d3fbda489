"""Fixture: a burned-in bench model (found by tests/test_gpu_flow_parity.py, round 5) whose Rayleigh phase velocity at T = 24 s
is EXACTLY the float32 S velocity of layer 20.  The reference's sregn96 divides by that layer's vertical wavenumber (zero) and
returns NaN for every kernel of that period (flag True, finite phase velocities).  Written with the COMPILED reference
(oracle/_ref, built by oracle/Makefile from /root/reference/src/SWD):  python tests/golden/make_exact_equality.py <model.npz>"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O
x = np.load(sys.argv[1])["x"][0]
n = 30
t = np.linspace(5, 44, 40)
vs, thk = x[:n], x[n:]
vp, rho, _, _ = O.empirical_relation(vs)
assert O.ref_available()
c, ka, kb, kr, kh, flag = O.ref_libsurf().adjoint_kernel(thk, vp, vs, rho, t, "Rc")
nanrow = np.nonzero(~np.isfinite(kb).all(axis=1))[0]
print("flag", flag, "NaN rows", nanrow, "c there", c[nanrow], "vs32 of layer 20", float(np.float32(vs[20])))
assert flag and list(nanrow) == [19] and c[19] == float(np.float32(vs[20]))
np.savez(os.path.join(ROOT, "tests", "golden", "exact_equality_reference.npz"), x=x, t=t, c=c, ka=ka, kb=kb, kr=kr, kh=kh,
         flag=np.array(flag), nan_row=nanrow)

"""One-off soak: Rayleigh phase velocities of the bench's own models (8192 chains x 40 periods by default), device
vs the C restatement of surfdisp96 (oracle/, itself bit-exact against the compiled reference): counts differing roots."""
import sys, time, ctypes; sys.path.insert(0, '.')
import numpy as np, torch
from bench import make_models, N_LAYER, NPER
from oracle import oracle as O
O.build(ref=False)
from rfsurfhmc_amd.model.model_surf import SurfWD
nchain = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 991206
t = np.linspace(5, 44, NPER)
xs = make_models(nchain, seed)
m = SurfWD(tRc=t)
d, flag = m.forward(xs)
lib = ctypes.CDLL('oracle/liboracle.so')
FP = ctypes.POINTER(ctypes.c_float); DP = ctypes.POINTER(ctypes.c_double)
f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
ndiff = nflag = 0; worst = 0.0
t0 = time.time()
for i in range(nchain):
    vs = xs[i, :N_LAYER]; thk = xs[i, N_LAYER:].copy()
    vp, rho, _, _ = O.empirical_relation(vs)
    a = [f32(thk), f32(vp), f32(vs), f32(rho)]
    cg = np.zeros(NPER); nsec = ctypes.c_long(0)
    ierr = lib.orc_surfdisp_rc(*[v.ctypes.data_as(FP) for v in a], N_LAYER, t.ctypes.data_as(DP), cg.ctypes.data_as(DP), NPER, ctypes.byref(nsec))
    ok = ierr != 1
    if ok != bool(flag[i]): nflag += 1; continue
    if not ok: continue
    bad = d[i] != cg
    if bad.any():
        ndiff += int(bad.sum()); worst = max(worst, float(np.abs(d[i] - cg)[bad].max() / cg[bad].max()))
print(f"{nchain} chains x {NPER} periods: {ndiff} differing roots, worst rel {worst:.2e}, flag mismatches {nflag}, oracle {time.time()-t0:.1f} s")

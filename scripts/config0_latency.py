"""BASELINE configs[0] shape on the device: SWD-only plugin, 10 layers, 36 Rc + 36 Rg periods; time per evaluation for few chains."""
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from rfsurfhmc_amd.model.model_surf import SurfWD
thk = np.array([3, 3, 4, 5, 5, 6, 7, 8, 10, 0.]); vs = np.linspace(2.9, 4.6, 10)
t = np.arange(5., 41.)
m = SurfWD(tRc=t, tRg=t)
x0 = np.hstack((vs, thk))
d, flag = m.forward(x0); m.set_obsdata(d * 1.01)
for nchain in [int(a) for a in sys.argv[1:]] or [1, 64, 512]:
    rng = np.random.default_rng(1)
    xs = np.tile(x0, (nchain, 1)) * (1 + 0.01 * rng.standard_normal((nchain, 20)))
    xs[:, :10] = np.sort(xs[:, :10], axis=1); xs[:, -1] = 0
    for _ in range(5): out = m.misfit_and_grad(xs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): out = m.misfit_and_grad(xs)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    print(f"{nchain:5d} chains: {ms:.3f} ms per evaluation (host arrays in and out), {nchain / ms * 1e3:.0f} evals/s, flags ok {bool(np.all(out[3]))}")

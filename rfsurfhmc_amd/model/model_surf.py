"""SurfWD -- host-side mirror of the reference plugin model/model_surf.py (same constructor,
``init(**kargs)`` keys, ``set_obsdata``, ``set_thk``, ``forward``, ``misfit``, ``misfit_and_grad``),
evaluated on the GPU through the fused B2 entry points.  Rayleigh blocks (tRc, tRg) only."""
import numpy as np

from ._plugin import FusedPlugin


class SurfWD(FusedPlugin):
    def __init__(self, mode=0, sphere=False, tRc=None, tRg=None, tLc=None, tLg=None, device=0):
        if mode != 0 or sphere:
            raise NotImplementedError("fundamental mode, flat earth only")
        if (tLc is not None and len(tLc) > 0) or (tLg is not None and len(tLg) > 0):
            raise NotImplementedError("Love-wave data are out of scope")
        self.mode, self.sphere, self.device = mode, sphere, device
        self.tRc = np.asarray(tRc, dtype=float) if tRc is not None and len(tRc) > 0 else None
        self.tRg = np.asarray(tRg, dtype=float) if tRg is not None and len(tRg) > 0 else None
        self.tLc = self.tLg = None
        self.ntRc = 0 if self.tRc is None else len(self.tRc)
        self.ntRg = 0 if self.tRg is None else len(self.tRg)
        self.ntLc = self.ntLg = 0
        self.nt = self.ntRc + self.ntRg

    @classmethod
    def init(cls, **kargs):
        """model_surf.py:31-38: keys tRc, tRg, tLc, tLg of param.yaml's swd block."""
        return cls(tRc=kargs["tRc"], tRg=kargs["tRg"], tLc=kargs.get("tLc"), tLg=kargs.get("tLg"))

    def _periods(self):
        return self.tRc, self.tRg

    def set_obsdata(self, dobs):
        self.dobs = dobs

    def set_thk(self, thk):
        self.thk = np.asarray(thk) * 1.0

    def forward(self, x):
        """(d[nt], flag) -- model_surf.py:81-133 (every block computed at tRc, as the reference does)."""
        single, dsyn, flag = self._forward(x, quirk=True)
        return (dsyn[0], bool(flag[0])) if single else (dsyn, flag)

    def misfit(self, x):
        d, flag = self.forward(x)
        if np.ndim(flag) == 0:
            return (0.5 * np.sum((d - self.dobs) ** 2), True) if flag else (0.0, flag)
        return np.where(flag, 0.5 * np.sum((d - self.dobs) ** 2, axis=1), 0.0), flag

    def misfit_and_grad(self, x):
        """(misfit, grad[2n], d[nt], flag) -- model_surf.py:155-228."""
        single, misfit, grad, dsyn, flag = self._eval(x)
        if single:
            if not flag[0]:      # reference failure return: (0.0, zeros(n), zeros(nt), False), :181-182
                return 0.0, np.zeros(grad.shape[1] // 2), dsyn[0], False
            return float(misfit[0]), grad[0], dsyn[0], True
        return misfit, grad, dsyn, flag

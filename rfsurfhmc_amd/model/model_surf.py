"""SurfWD -- host-side mirror of the reference plugin model/model_surf.py (same constructor,
``init(**kargs)`` keys, ``set_obsdata``, ``set_thk``, ``forward``, ``misfit``, ``misfit_and_grad``),
evaluated on the GPU through the fused B2 entry points.  All four blocks (tRc, tRg, tLc, tLg),
flat or spherical earth, fundamental or higher modes (``mode``; the warm-started root search of the trajectory entries
covers the fundamental only, higher modes always go through the reference-semantics search).

Reference behaviour kept by default (``reference_periods=True``): forward() evaluates EVERY block at
tRc (model_surf.py:104-131) and misfit_and_grad() evaluates the Lc and Lg blocks at tRc
(model_surf.py:199-216), so those blocks need len(tRc) rows -- the reference raises otherwise, and so
does this class.  ``reference_periods=False`` evaluates every block at its own periods."""
import numpy as np

from ._plugin import FusedPlugin


class SurfWD(FusedPlugin):
    def __init__(self, mode=0, sphere=False, tRc=None, tRg=None, tLc=None, tLg=None, device=0,
                 reference_periods=True):
        if int(mode) != mode or mode < 0:
            raise ValueError("mode should be a non-negative integer (0 = fundamental)")
        self.mode, self.sphere, self.device = int(mode), bool(sphere), device
        self.reference_periods = reference_periods
        for name, t in (("tRc", tRc), ("tRg", tRg), ("tLc", tLc), ("tLg", tLg)):
            arr = np.asarray(t, dtype=float) if t is not None and len(t) > 0 else None
            setattr(self, name, arr)
            setattr(self, "n" + name, 0 if arr is None else len(arr))
        self.nt = self.ntRc + self.ntRg + self.ntLc + self.ntLg

    @classmethod
    def init(cls, **kargs):
        """model_surf.py:31-38: keys tRc, tRg, tLc, tLg of param.yaml's swd block."""
        return cls(tRc=kargs["tRc"], tRg=kargs["tRg"], tLc=kargs.get("tLc"), tLg=kargs.get("tLg"))

    def _love_eval_periods(self, t, name):
        if t is None or not self.reference_periods:
            return t
        if self.tRc is None:     # the reference hands None to the pybind11 binding here
            raise TypeError(f"the reference evaluates its {name} block at tRc, which is not set "
                            "(pass reference_periods=False to use the block's own periods)")
        if len(t) != self.ntRc:  # d[k1:k2] = cg with len(cg) == len(tRc)
            raise ValueError(f"could not broadcast input array from shape ({self.ntRc},) into shape ({len(t)},)")
        return self.tRc

    def _swd_config(self):
        return (self.tRc, self.tRg, self._love_eval_periods(self.tLc, "Lc"),
                self._love_eval_periods(self.tLg, "Lg"), self.sphere)

    def _swd_mode(self):
        return self.mode

    def set_obsdata(self, dobs):
        self.dobs = dobs

    def set_thk(self, thk):
        self.thk = np.asarray(thk) * 1.0

    def forward(self, x):
        """(d[nt], flag) -- model_surf.py:81-133 (every block computed at tRc, as the reference does)."""
        single, dsyn, flag = self._forward(x, quirk=self.reference_periods)
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

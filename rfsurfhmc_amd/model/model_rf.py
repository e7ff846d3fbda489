"""ReceiverFunc -- host-side mirror of the reference plugin model/model_rf.py."""
import numpy as np

from .._lib import RfParams
from ._plugin import FusedPlugin


class ReceiverFunc(FusedPlugin):
    def __init__(self, ray_p, nt, dt, gauss, time_shift, water_level=0.001, type_="p", method="time", device=0):
        self.ray_p, self.nt, self.dt, self.gauss = ray_p, nt, dt, gauss
        self.time_shift, self.water_level, self.rf_type, self.method = time_shift, water_level, type_, method
        self.t = np.arange(nt) * dt - time_shift
        self.device = device

    @classmethod
    def init(cls, **kargs):
        """model_rf.py:20-30: keys of param.yaml's rf block."""
        return cls(kargs["ray_p"], kargs["nt"], kargs["dt"], kargs["gauss"], kargs["time_shift"],
                   kargs["water_level"], kargs["type"], kargs["method"])

    def _rf_params(self):
        if self.rf_type in ("P", "p"):
            irf = 1
        elif self.rf_type in ("S", "s"):
            irf = 2
        else:
            raise ValueError("rf_type should be one of [P,p,S,s]")
        return RfParams(float(self.ray_p), int(self.nt), float(self.dt), float(self.gauss),
                        float(self.time_shift), float(self.water_level), irf,
                        0 if self.method == "time" else 1)      # src/RF/main.cpp:43,115: anything but "time" is "freq"

    def set_obsdata(self, dobs):
        self.dobs = dobs

    def set_thk(self, thk):
        self.thk = np.asarray(thk) * 1.0

    def forward(self, x):
        """rf[nt] -- model_rf.py:79-116."""
        single, dsyn, _ = self._forward(x)
        return dsyn[0] if single else dsyn

    def misfit(self, x):
        d = self.forward(x)
        return 0.5 * np.sum((d - self.dobs) ** 2, axis=-1)

    def misfit_and_grad(self, x):
        """(misfit, grad[2n], d[nt]) -- model_rf.py:137-198 (3-tuple, no flag)."""
        single, misfit, grad, dsyn, _ = self._eval(x)
        if single:
            return float(misfit[0]), grad[0], dsyn[0]
        return misfit, grad, dsyn

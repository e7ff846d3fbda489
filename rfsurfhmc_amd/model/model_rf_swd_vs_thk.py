"""Joint_RF_SWD -- host-side mirror of the reference plugin model/model_rf_swd_vs_thk.py."""
import numpy as np

from ._plugin import FusedPlugin
from .model_rf import ReceiverFunc
from .model_surf import SurfWD


class Joint_RF_SWD(FusedPlugin):
    def __init__(self, sigma1, sigma2, rfmodel: ReceiverFunc, swdmodel: SurfWD, device=None):
        self.sigma1, self.sigma2 = sigma1, sigma2
        self.rfmodel, self.swdmodel = rfmodel, swdmodel
        self.ndata = rfmodel.nt + swdmodel.nt
        self.device = rfmodel.device if device is None else device

    def _rf_params(self):
        return self.rfmodel._rf_params()

    def _swd_config(self):
        return self.swdmodel._swd_config()

    def _swd_mode(self):
        return self.swdmodel._swd_mode()

    def _sigmas(self):
        return self.sigma1, self.sigma2

    def set_obsdata(self, rfobs, swdobs):
        """model_rf_swd_vs_thk.py:14-25."""
        self.rfobs = np.asarray(rfobs) * 1.0
        self.swdobs = np.asarray(swdobs) * 1.0
        self.rfmodel.set_obsdata(self.rfobs)
        self.swdmodel.set_obsdata(self.swdobs)
        self.dobs = np.concatenate((self.rfobs, self.swdobs))

    def forward(self, x):
        """(drf, dswd, flag) -- model_rf_swd_vs_thk.py:27-49."""
        single, dsyn, flag = self._forward(x, quirk=self.swdmodel.reference_periods)
        n1 = self.rfmodel.nt
        if single:
            return dsyn[0, :n1], dsyn[0, n1:], bool(flag[0])
        return dsyn[:, :n1], dsyn[:, n1:], flag

    def misfit(self, x):
        drf, dswd, flag = self.forward(x)
        wt = (self.sigma1 / self.sigma2) ** 2 * self.rfmodel.nt / self.swdmodel.nt
        m = 0.5 * np.sum((drf - self.rfobs) ** 2, axis=-1) + 0.5 * np.sum((dswd - self.swdobs) ** 2, axis=-1) * wt
        if np.ndim(flag) == 0:
            return (m, True) if flag else (0.0, flag)
        return np.where(flag, m, 0.0), flag

    def misfit_and_grad(self, x):
        """(misfit, grad[2n], dsyn[ndata], flag) -- model_rf_swd_vs_thk.py:66-86."""
        single, misfit, grad, dsyn, flag = self._eval(x)
        if single:
            return float(misfit[0]), grad[0], dsyn[0], bool(flag[0])
        return misfit, grad, dsyn, flag

"""Forward smoke run in the role of the reference's test_forward.py: time-domain P receiver function + 36 Rc + 36 Rg
periods for the 7-layer model, through the plugin classes (GPU).  Writes syn_test.npz (and syn_test.png when
matplotlib is installed).  Run from the repo root: python examples/forward_smoke.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from rfsurfhmc_amd.model.model_rf import ReceiverFunc
from rfsurfhmc_amd.model.model_rf_swd_vs_thk import Joint_RF_SWD
from rfsurfhmc_amd.model.model_surf import SurfWD


def main():
    tRc = np.linspace(5, 40, 36)
    tRg = tRc.copy()
    model_swd = SurfWD(tRc=tRc, tRg=tRg)
    rayp, dt, nt, gauss, time_shift, water = 0.045, 0.1, 500, 1.0, 5.0, 0.001
    model_rf = ReceiverFunc(rayp, nt, dt, gauss, time_shift, water, "P", "time")
    thk = np.array([6, 6, 13, 5, 10, 30, 0.0])
    vs = np.array([3.2, 3.4, 3.46, 3.7, 3.9, 4.5, 4.7])
    model_swd.set_thk(thk)
    model_rf.set_thk(thk)
    model = Joint_RF_SWD(1.0, 1.0, model_rf, model_swd)
    drsyn, dssyn, flag = model.forward(np.hstack((vs, thk)))
    t_rf = np.linspace(0, (nt - 1) * dt, nt) - time_shift
    np.savez("syn_test.npz", t_rf=t_rf, rf_syn=drsyn, tRc=tRc, Rc_syn=dssyn[:len(tRc)], tRg=tRg, Rg_syn=dssyn[len(tRc):])
    print("flag", flag, "rf peak %.4f at %.1f s" % (drsyn.max(), t_rf[np.argmax(drsyn)]),
          "Rc %.4f..%.4f" % (dssyn[0], dssyn[len(tRc) - 1]), "Rg %.4f..%.4f" % (dssyn[len(tRc)], dssyn[-1]))
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
    except ImportError:
        return
    plt.figure(1, figsize=(14, 30))
    for k, (xx, yy, title) in enumerate([(t_rf, drsyn, "rf_syn"), (tRc, dssyn[:len(tRc)], "Rc_syn"),
                                         (tRg, dssyn[len(tRc):], "Rg_syn")]):
        plt.subplot(3, 1, k + 1)
        plt.plot(xx, yy)
        plt.title(title)
    plt.savefig("./syn_test.png")


if __name__ == "__main__":
    tic = time.time()
    main()
    print("time elapse: {}".format(time.time() - tic))

"""Shared machinery of the three model plugins: one rfs_ctx configured with the plugin's data
(rfs_joint_setup) and batched misfit_and_grad / forward over it."""
import ctypes

import numpy as np

from .._lib import Context, RfParams, SwdParams, hptr


class FusedPlugin:
    """x may be 1-D [2n] (reference call shape: scalars / 1-D arrays come back) or 2-D [nchain, 2n]."""

    _ctx = None
    device = 0
    _warm_start = None       # None = the library's default (rfs_set_option "swd_warm_start": 1)

    def _rf_params(self):
        return None

    def _swd_config(self):
        """(tRc, tRg, tLc, tLg, sphere): the periods each SWD block is evaluated at in misfit_and_grad."""
        return None, None, None, None, False

    def _swd_mode(self):
        """libsurf's ``mode`` (0 = fundamental, k > 0 = k-th higher mode; model_surf.py:5-7)."""
        return 0

    def _sigmas(self):
        return 1.0, 1.0

    def _configure(self, nlayer, dobs):
        if self._ctx is None:
            self._ctx = Context(device=self.device, max_chains=1 << 20, max_layers=128)
            if self._warm_start is not None:
                self._ctx.set_option("swd_warm_start", self._warm_start)
        ctx = self._ctx
        rf = self._rf_params()
        *tw, sphere = self._swd_config()
        tw = [np.ascontiguousarray(t, dtype=np.float64) if t is not None else np.zeros(0) for t in tw]
        s1, s2 = self._sigmas()
        d = None if dobs is None else np.ascontiguousarray(dobs, dtype=np.float64)
        swd = None
        if sum(len(t) for t in tw) > 0:
            swd = SwdParams(*[len(t) for t in tw], *[t.ctypes.data if len(t) else None for t in tw],
                            int(bool(sphere)), int(self._swd_mode()))
        ctx.check(ctx.L.rfs_joint_setup2(ctx.h, int(nlayer), ctypes.byref(rf) if rf is not None else None,
                                         ctypes.byref(swd) if swd is not None else None,
                                         float(s1), float(s2), hptr(d) if d is not None else None))
        tRc, tRg = tw[0], tw[1]
        self._cfg = (int(nlayer), None if d is None else d.tobytes())
        self._keep = (tw, d)

    def set_warm_start(self, mode):
        """Root search inside trajectories (include/rfsurf.h, option "swd_warm_start"): 0 = every evaluation by the
        reference-semantics search (bit-exact float32 roots, bit-reproducible whatever the schedule); 1 = the
        trajectory entries continue the previous step's roots (default: ~1.5x the leapfrog rate, roots within the
        reference's own 1e-6 c refinement tolerance); 2 = also misfit_and_grad / misfit_and_grad_device (the caller
        promises consecutive calls are consecutive models of the same chains)."""
        self._warm_start = int(mode)
        if self._ctx is not None:
            self._ctx.set_option("swd_warm_start", self._warm_start)

    def reset_warm_start(self):
        """The next evaluation goes through the full search for every chain (samplers call this at checkpoints)."""
        if self._ctx is not None:
            self._ctx.set_option("swd_warm_reset", 1)

    def _ensure(self, nlayer):
        dobs = getattr(self, "dobs", None)
        key = (int(nlayer), None if dobs is None else np.ascontiguousarray(dobs, dtype=np.float64).tobytes())
        if getattr(self, "_cfg", None) != key:
            self._configure(nlayer, dobs)
        return self._ctx

    def _eval(self, x):
        x = np.ascontiguousarray(np.asarray(x, dtype=np.float64))
        single = x.ndim == 1
        x2 = np.atleast_2d(x)
        nchain, nx = x2.shape
        ctx = self._ensure(nx // 2)
        nd = ctx.L.rfs_ndata(ctx.h)
        misfit = np.zeros(nchain); grad = np.zeros((nchain, nx)); dsyn = np.zeros((nchain, nd))
        flag = np.zeros(nchain, dtype=np.int32)
        ctx.check(ctx.L.rfs_joint_misfit_grad(ctx.h, nchain, hptr(x2), hptr(misfit), hptr(grad), hptr(dsyn),
                                              hptr(flag)))
        return single, misfit, grad, dsyn, flag.astype(bool)

    def _forward(self, x, quirk=True):
        x = np.ascontiguousarray(np.asarray(x, dtype=np.float64))
        single = x.ndim == 1
        x2 = np.atleast_2d(x)
        nchain, nx = x2.shape
        ctx = self._ensure(nx // 2)
        nd = ctx.L.rfs_ndata(ctx.h)
        dsyn = np.zeros((nchain, nd)); flag = np.zeros(nchain, dtype=np.int32)
        ctx.check(ctx.L.rfs_joint_forward(ctx.h, nchain, hptr(x2), int(quirk), hptr(dsyn), hptr(flag)))
        return single, dsyn, flag.astype(bool)

    # ---- device-resident path (torch CUDA tensors in, torch tensors out; no host copies) ----
    def misfit_and_grad_device(self, x):
        """x: torch.float64 CUDA tensor [nchain, 2n] -> (misfit, grad, dsyn, flag) CUDA tensors."""
        import torch
        assert x.is_cuda and x.dtype == torch.float64 and x.dim() == 2 and x.is_contiguous()
        nchain, nx = x.shape
        ctx = self._ensure(nx // 2)
        ctx.check(ctx.L.rfs_set_stream(ctx.h, ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)))
        nd = ctx.L.rfs_ndata(ctx.h)
        misfit = torch.empty(nchain, dtype=torch.float64, device=x.device)
        grad = torch.empty(nchain, nx, dtype=torch.float64, device=x.device)
        dsyn = torch.empty(nchain, nd, dtype=torch.float64, device=x.device)
        flag = torch.empty(nchain, dtype=torch.int32, device=x.device)
        ctx.check(ctx.L.rfs_joint_misfit_grad_dev(ctx.h, nchain, x.data_ptr(), misfit.data_ptr(), grad.data_ptr(),
                                                  dsyn.data_ptr(), flag.data_ptr()))
        return misfit, grad, dsyn, flag

    def leapfrog_device(self, x0, p0, dt, L, bounds, sort_by_length=None):
        """Device-resident leapfrog trajectories (pyhmc/hmc.py:140-190) for all chains at once.

        x0, p0: float64 CUDA [nchain, 2n]; dt: float64 CUDA [nchain]; L: int32 CUDA [nchain];
        bounds: float64 CUDA [2n, 2].  Returns dict(xnew, Ucur, Unew, Hcur, Hnew, dsyn_cur, dsyn_new, ok).

        sort_by_length: run the chains in order of decreasing L so that leapfrog step s only evaluates the chains
        with L > s (rfs_leapfrog_dev2); every chain's result is the same as in the unsorted schedule.  Default
        (None): on from 16384 chains -- below that one evaluation costs the same ~10 ms whatever the number of
        chains (the root search is latency bound), so dropping finished chains buys nothing; measured at 32768
        chains x 30 layers with L drawn in [5, 20]: 503 k -> 739 k chain-steps/s."""
        import torch
        nchain, nx = x0.shape
        ctx = self._ensure(nx // 2)
        dev = x0.device
        ctx.check(ctx.L.rfs_set_stream(ctx.h, ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        nd = ctx.L.rfs_ndata(ctx.h)
        f64 = dict(dtype=torch.float64, device=dev)
        out = dict(xnew=torch.empty(nchain, nx, **f64), Ucur=torch.empty(nchain, **f64),
                   Unew=torch.empty(nchain, **f64), Hcur=torch.empty(nchain, **f64),
                   Hnew=torch.full((nchain,), float("inf"), **f64), dsyn_cur=torch.empty(nchain, nd, **f64),
                   dsyn_new=torch.empty(nchain, nd, **f64), ok=torch.empty(nchain, dtype=torch.int32, device=dev))
        Lh = L.cpu().numpy()
        if Lh.min() < 1:
            raise ValueError("leapfrog_device: every chain needs L >= 1")
        Lmax = int(Lh.max())
        order = None
        nactive = None
        if sort_by_length is None:
            sort_by_length = nchain >= 16384
        if sort_by_length and nchain > 1 and Lh.min() != Lmax:
            oh = np.argsort(-Lh, kind="stable")
            order = torch.from_numpy(oh).to(dev)
            x0, p0, dt, L = (t.index_select(0, order).contiguous() for t in (x0, p0, dt, L))
            # nactive[s] = #{chains with L > s}: a histogram of L and a suffix sum (not an [Lmax, nchain] matrix)
            hist = np.bincount(Lh, minlength=Lmax + 1)
            nactive = np.ascontiguousarray((nchain - np.cumsum(hist)[:Lmax]).astype(np.int32))
        ctx.check(ctx.L.rfs_leapfrog_dev2(ctx.h, nchain, x0.data_ptr(), p0.data_ptr(), dt.data_ptr(), L.data_ptr(),
                                          Lmax, hptr(nactive) if nactive is not None else None, bounds.data_ptr(),
                                          out["xnew"].data_ptr(), out["Ucur"].data_ptr(),
                                          out["Unew"].data_ptr(), out["Hcur"].data_ptr(), out["Hnew"].data_ptr(),
                                          out["dsyn_cur"].data_ptr(), out["dsyn_new"].data_ptr(), out["ok"].data_ptr()))
        if order is not None:
            inv = torch.empty_like(order)
            inv[order] = torch.arange(nchain, device=dev)
            out = {k: v.index_select(0, inv) for k, v in out.items()}
        return out

    def set_inverse_mass(self, minv, nlayer=None):
        """Diagonal inverse mass of the device leapfrog (rfs_set_inverse_mass): minv[2n] > 0, or None for the
        reference's identity.  Has to be set again after set_obsdata / a change of layer count (the context is
        reconfigured then)."""
        if minv is None:
            self._minv = None
            if self._ctx is not None and getattr(self, "_cfg", None) is not None:
                self._ctx.check(self._ctx.L.rfs_set_inverse_mass(self._ctx.h, None))
            return
        m = np.ascontiguousarray(np.asarray(minv, dtype=np.float64))
        ctx = self._ensure(len(m) // 2 if nlayer is None else nlayer)
        ctx.check(ctx.L.rfs_set_inverse_mass(ctx.h, hptr(m)))
        self._minv = m

    def flow_step(self, st):
        """One call of rfs_flow_step on the state dict ``st`` (CUDA tensors x, p, dt, rem, fresh, bounds, Ucur, Hcur,
        Unew, Hnew, dsyn_cur, dsyn_new, ok, done -- see include/rfsurf.h): one evaluation per chain, every chain at
        its own point of its own trajectory."""
        import torch
        x = st["x"]
        nchain, nx = x.shape
        ctx = self._ensure(nx // 2)
        ctx.check(ctx.L.rfs_set_stream(ctx.h, ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)))
        args = [st[k].data_ptr() for k in ("x", "p", "dt", "rem", "fresh", "bounds", "Ucur", "Hcur", "Unew", "Hnew",
                                           "dsyn_cur", "dsyn_new", "ok", "done")]
        nxt = self._flow_next(st)
        if st.get("rec") is not None and hasattr(ctx.L, "rfs_flow_step3"):
            # records of the chains that complete a trajectory in this call, straight into the caller's pinned buffer
            # (rfs_flow_records): st["rec"] = (pinned float64 tensor used as a ring, its slots, want_dsyn, this call's stamp, reset)
            from .._lib import FlowRecords
            buf, cap, want, stamp, reset = st["rec"]
            rec = FlowRecords(buf.data_ptr(), buf.numel() * buf.element_size(), int(cap), int(bool(want)), float(stamp), int(bool(reset)))
            ctx.check(ctx.L.rfs_flow_step3(ctx.h, nchain, *args, ctypes.byref(nxt) if nxt is not None else None, ctypes.byref(rec)))
        elif nxt is not None:       # restarts on the device (rfs_flow_step2), state from flow_restart_state()
            ctx.check(ctx.L.rfs_flow_step2(ctx.h, nchain, *args, ctypes.byref(nxt)))
        else:
            ctx.check(ctx.L.rfs_flow_step(ctx.h, nchain, *args))

    @staticmethod
    def _flow_next(st):
        if "nxt_have" not in st:
            return None
        from .._lib import FlowNext
        return FlowNext(*[st[k].data_ptr() if st.get(k) is not None else None for k in
                          ("nxt_have", "nxt_u", "nxt_p", "nxt_rem", "xstart", "res_x", "res_val", "res_dsyn", "gsave", "kick")])

    def flow_deposit(self, st, stream, buf, n, o_idx, o_u, o_p, o_rem):
        """rfs_flow_deposit on the state ``st``: the n listed chains' acceptance draws, next momenta and lengths (o_rem None:
        deferred form) out of ONE byte buffer ``buf`` -- device memory, or pinned host memory the launch reads over the link --
        at the given byte offsets, in one launch on ``stream`` (a torch stream)."""
        nchain, nx = st["x"].shape
        ctx = self._ensure(nx // 2)
        base = buf.data_ptr()
        P = lambda off: None if off is None else base + off
        nxt = self._flow_next(st)
        ctx.check(ctx.L.rfs_flow_deposit(ctx.h, ctypes.c_void_p(stream.cuda_stream), nchain, int(n), P(o_idx), P(o_u), P(o_p), P(o_rem),
                                         ctypes.byref(nxt)))

    def flow_restart(self, st, buf, n1, o_idx1, o_xkeep, n2, o_idx2, o_p, o_rem, o_dt, n3, o_idx3):
        """rfs_flow_restart on the state ``st``: the lists and rows live in ONE byte buffer ``buf`` at the given byte offsets
        (None: absent) -- a uint8 CUDA tensor the caller copied up in one piece, or a PINNED host tensor, which the launch
        reads over the link (pinned host memory is mapped into the device's address space: right for a few KB; the caller
        copies anything above ~64 KB to the device first)."""
        import torch
        x = st["x"]
        nchain, nx = x.shape
        ctx = self._ensure(nx // 2)
        ctx.check(ctx.L.rfs_set_stream(ctx.h, ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)))
        base = buf.data_ptr()
        P = lambda off: None if off is None else base + off
        nh = st.get("nxt_have")
        ctx.check(ctx.L.rfs_flow_restart(ctx.h, nchain, n1, P(o_idx1), P(o_xkeep), n2, P(o_idx2), P(o_p), P(o_rem), P(o_dt),
                                         n3, P(o_idx3), st["x"].data_ptr(), st["p"].data_ptr(), st["rem"].data_ptr(),
                                         st["dt"].data_ptr(), st["fresh"].data_ptr(), st["ok"].data_ptr(),
                                         nh.data_ptr() if nh is not None else None))

    def flow_restart_state(self, st, want_dsyn=False, deferred=False):
        """Adds to a flow state the arrays of rfs_flow_next: deposits for the next trajectory (nxt_have / nxt_u / nxt_p /
        nxt_rem), the start model of the running one (xstart) and the parked results of the last completed one (res_x,
        res_val = [Ucur, Hcur, Hnew, Unew], res_dsyn if wanted).  deferred: the form for samplers whose next step size
        depends on the finished trajectory (no nxt_rem; gsave / kick for the first half kick one call later)."""
        import torch
        x = st["x"]
        nchain, nx = x.shape
        f64 = dict(dtype=torch.float64, device=x.device)
        st.update(nxt_have=torch.zeros(nchain, dtype=torch.int32, device=x.device), nxt_u=torch.zeros(nchain, **f64),
                  nxt_p=torch.zeros(nchain, nx, **f64), nxt_rem=torch.zeros(nchain, dtype=torch.int32, device=x.device),
                  xstart=x.clone(), res_x=torch.zeros(nchain, nx, **f64), res_val=torch.zeros(nchain, 4, **f64),
                  res_dsyn=torch.zeros_like(st["dsyn_new"]) if want_dsyn else None, gsave=None, kick=None)
        if deferred:
            st.update(nxt_rem=None, gsave=torch.zeros(nchain, nx, **f64),
                      kick=torch.zeros(nchain, dtype=torch.int32, device=x.device))
        return st

    def flow_state(self, x0, dt, bounds):
        """Fresh state for flow_step: x0 float64 CUDA [nchain, 2n], dt float64 CUDA [nchain], bounds [2n, 2]."""
        import torch
        nchain, nx = x0.shape
        ctx = self._ensure(nx // 2)
        nd = ctx.L.rfs_ndata(ctx.h)
        dev = x0.device
        f64 = dict(dtype=torch.float64, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        return dict(x=x0.clone().contiguous(), p=torch.zeros(nchain, nx, **f64), dt=dt.contiguous(),
                    rem=torch.full((nchain,), -1, **i32), fresh=torch.zeros(nchain, **i32),
                    bounds=bounds.contiguous(), Ucur=torch.zeros(nchain, **f64), Hcur=torch.zeros(nchain, **f64),
                    Unew=torch.zeros(nchain, **f64), Hnew=torch.zeros(nchain, **f64),
                    dsyn_cur=torch.zeros(nchain, nd, **f64), dsyn_new=torch.zeros(nchain, nd, **f64),
                    ok=torch.ones(nchain, **i32), done=torch.zeros(nchain, **i32))


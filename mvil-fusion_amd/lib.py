"""Loader + thin Python calling layer over the C-ABI (include/vilsolve.h).

`Backend` drives any shared library that exports the vilsolve entry points under a prefix:
  * prefix "vil_"  -> csrc/libvilsolve.so, the HIP product (needs a GPU; fails loudly without one)
  * prefix "orc_"  -> oracle/liboracle.so, the CPU restatement -- constructed ONLY by tests/,
                      __graft_entry__.smoke() and bench.py's cpu_baseline leg.
There is no CPU fallback inside the product path: if libvilsolve.so is missing or no HIP device is
present, `load_vilsolve()` / `Backend.create()` raise.
"""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VIL_LIB") or os.path.join(_HERE, "csrc", "libvilsolve.so")      # VIL_LIB: tooling builds (tools/probe_step.py), never a fallback

STATUS = {0: "ok", -1: "invalid argument", -2: "device error", -3: "non-finite", -4: "not positive definite", -5: "comm error", -6: "unsupported"}


class VilError(RuntimeError):
    def __init__(self, what, status):
        super().__init__("%s failed: status %d (%s)" % (what, status, STATUS.get(status, "?")))
        self.status = status


def load_vilsolve(path=LIB_PATH):
    if not os.path.exists(path):
        raise RuntimeError("HIP extension %s is missing -- run `python -c 'import __graft_entry__ as g; g.build()'`; there is no CPU fallback" % path)
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


_dp = C.POINTER(C.c_double)


def _ptr(a):
    return a.ctypes.data_as(_dp)


class Backend:
    def __init__(self, cdll, prefix, device=0, rank=0, world=1):
        self.lib, self.prefix = cdll, prefix
        self.ctx = None
        self.has_ctx = prefix == "vil_"
        if self.has_ctx:
            cfg = abi.VilDeviceCfg(device, rank, world, 0)
            ctx = C.c_void_p()
            f = self.lib.vil_create
            f.restype = C.c_int
            st = f(C.byref(cfg), C.byref(ctx))
            if st != 0:
                raise VilError("vil_create", st)
            self.ctx = ctx

    def close(self):
        if self.has_ctx and self.ctx is not None:
            self.lib.vil_destroy.restype = None
            self.lib.vil_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _call(self, name, *args):
        f = getattr(self.lib, self.prefix + name)
        f.restype = C.c_int
        if self.has_ctx:
            args = (self.ctx,) + args
        st = f(*args)
        if st != 0:
            raise VilError(self.prefix + name, st)
        return st

    # -- Evaluate()-compatible per-class residuals and Jacobians
    def eval_factors(self, w, cls, jac=True):
        nr, nj = w.eval_sizes(cls)
        r = np.zeros(max(nr, 1)); J = np.zeros(max(nj, 1))
        p, s = w.c_problem(), w.c_state()
        self._call("eval_factors", C.byref(p), C.byref(s), C.c_int(cls), _ptr(r), _ptr(J) if jac else C.cast(None, _dp))
        return r[:nr], J[:nj]

    def linearize(self, w, opts=None):
        opts = opts or abi.default_options()
        D = w.D
        cost = C.c_double(0.0); S = np.zeros((D, D)); g = np.zeros(D)
        p, s = w.c_problem(), w.c_state()
        self._call("linearize", C.byref(p), C.byref(s), C.byref(opts), C.byref(cost), _ptr(S), _ptr(g))
        return cost.value, S, g

    def solve(self, w, opts=None):
        """In place on w's state arrays (like Estimator::optimization() on para_*). Returns VilSummary."""
        opts = opts or abi.default_options()
        p, s = w.c_problem(), w.c_state()
        summ = abi.VilSummary()
        self._call("solve", C.byref(p), C.byref(s), C.byref(opts), C.byref(summ))
        return summ

    def marginalize(self, w, flag=abi.MARGIN_OLD, icp_marg=-1, lps_marg=-1, opts=None, threads=4):
        opts = opts or abi.default_options()
        p, s = w.c_problem(), w.c_state()
        spec = abi.VilMargSpec(flag, icp_marg, lps_marg, threads)
        out = abi.PriorOut(w.K)
        self._call("marginalize", C.byref(p), C.byref(s), C.byref(opts), C.byref(spec), C.byref(out.c))
        return out

    def gauge_fix(self, pose0_before, w):
        f = getattr(self.lib, self.prefix + "gauge_fix")
        f.restype = C.c_int
        s = w.c_state()
        p0 = abi.f64(pose0_before)
        st = f(_ptr(p0), C.byref(s))
        if st != 0:
            raise VilError("gauge_fix", st)

    # -- resident API (HIP library only)
    def upload(self, w):
        p, s = w.c_problem(), w.c_state()
        self._call("upload", C.byref(p), C.byref(s))

    def solve_resident(self, opts=None):
        opts = opts or abi.default_options()
        summ = abi.VilSummary()
        self._call("solve_resident", C.byref(opts), C.byref(summ))
        return summ

    def reset_state(self):
        self._call("reset_state")

    # -- window residency across frames (HIP library only; include/vilsolve.h)
    def lidar_reset(self):
        self._call("lidar_reset")

    def lidar_push(self, plane_const, edge_const):
        pc, ec = abi.f64(plane_const).reshape(-1, 7), abi.f64(edge_const).reshape(-1, 9)
        self._call("lidar_push", C.c_int32(len(pc)), _ptr(pc) if len(pc) else C.cast(None, _dp), C.c_int32(len(ec)), _ptr(ec) if len(ec) else C.cast(None, _dp))

    def lidar_drop(self, slab):
        self._call("lidar_drop", C.c_int32(slab))

    def set_gauge_fix(self, on=True):
        self._call("set_gauge_fix", C.c_int32(1 if on else 0))

    def marginalize_resident(self, w, flag=abi.MARGIN_OLD, icp_marg=-1, lps_marg=-1, opts=None):
        """Marginalise the window the last solve() left on the device, at its solved state (w only supplies that state for x0)."""
        opts = opts or abi.default_options()
        s = w.c_state()
        spec = abi.VilMargSpec(flag, icp_marg, lps_marg, 4)
        out = abi.PriorOut(w.K)
        self._call("marginalize_resident", C.byref(s), C.byref(opts), C.byref(spec), C.byref(out.c))
        return out

    # -- the fully resident window (HIP library only; include/vilsolve.h: vil_win_*)
    def win_open(self, K, max_tracks, max_samples, noise, G, sqrt_info_px, tr_over_row, q_lb, t_lb, use_td=1):
        cfg = abi.VilWinCfg()
        cfg.K, cfg.max_tracks, cfg.max_samples, cfg.use_td = int(K), int(max_tracks), int(max_samples), int(use_td)
        for k in range(4):
            cfg.noise[k] = float(noise[k]); cfg.q_lb[k] = float(q_lb[k])
        for k in range(3):
            cfg.G[k] = float(G[k]); cfg.t_lb[k] = float(t_lb[k])
        cfg.sqrt_info_px, cfg.tr_over_row = float(sqrt_info_px), float(tr_over_row)
        self._call("win_open", C.byref(cfg))

    def win_push_frame(self, fr):
        """fr: dict with dt (n), acc (n x 3), gyr (n x 3), acc0, gyr0, lin_ba, lin_bg, obs_track (m), obs (m x 8), plane (p x 7), edge (e x 9)."""
        f = self._wf = getattr(self, "_wf", None) or abi.VilWinFrame()
        c = np.ascontiguousarray
        keep = (c(fr["dt"], np.float64), c(fr["acc"], np.float64), c(fr["gyr"], np.float64), c(fr["obs_track"], np.int32), c(fr["obs"], np.float64), c(fr["plane"], np.float64), c(fr["edge"], np.float64))
        ad = lambda a: a.__array_interface__["data"][0]
        f.n_samples = len(keep[0]); f.dt, f.acc, f.gyr = ad(keep[0]), ad(keep[1]), ad(keep[2])
        f.acc0[:] = [float(v) for v in fr["acc0"]]; f.gyr0[:] = [float(v) for v in fr["gyr0"]]; f.lin_ba[:] = [float(v) for v in fr["lin_ba"]]; f.lin_bg[:] = [float(v) for v in fr["lin_bg"]]
        f.n_obs = len(keep[3]); f.obs_track, f.obs = ad(keep[3]), ad(keep[4])
        f.n_plane = keep[5].size // 7; f.plane_const = ad(keep[5]); f.n_edge = keep[6].size // 9; f.edge_const = ad(keep[6])
        self._call("win_push_frame", C.byref(f))

    def win_drop_frame(self, flag):
        self._call("win_drop_frame", C.c_int32(int(flag)))

    def win_solve(self, w, opts=None):
        """w: a Window whose visual structure is given per landmark (w.lm_track / lm_start / lm_nobs, int32); state in place, like solve()."""
        opts = opts or abi.default_options()
        p = self._wp = getattr(self, "_wp", None) or abi.VilWinProblem()
        d = abi._d
        p.L = w.L; p.lm_track, p.lm_start, p.lm_nobs = d(w.lm_track), d(w.lm_start), d(w.lm_nobs)
        p.lm_const, p.pose_const, p.sb_const = d(w.lm_const), d(w.pose_const), d(w.sb_const)
        p.ex_const, p.td_const = int(w.ex_const), int(w.td_const)
        p.n_icp = len(w.icp_ids); p.icp_ids, p.icp_const = d(w.icp_ids), d(w.icp_const)
        p.n_lps = len(w.lps_ids); p.lps_ids, p.lps_const = d(w.lps_ids), d(w.lps_const)
        s = abi.VilState()
        s.K, s.L = w.K, w.L
        s.pose, s.speedbias, s.ex_pose, s.td, s.inv_depth = d(w.pose), d(w.speedbias), d(w.ex_pose), d(w.td), d(w.inv_depth)
        summ = abi.VilSummary()
        self._call("win_solve", C.byref(p), C.byref(s), C.byref(opts), C.byref(summ))
        return summ

    def win_marginalize(self, flag=abi.MARGIN_OLD, icp_marg=-1, lps_marg=-1, opts=None):
        opts = opts or abi.default_options()
        spec = abi.VilMargSpec(flag, icp_marg, lps_marg, 4)
        info = abi.VilWinPriorInfo()
        self._call("win_marginalize", C.byref(opts), C.byref(spec), C.byref(info))
        return info

    def win_prior_download(self, K):
        out = abi.PriorOut(K)
        self._call("win_prior_download", C.byref(out.c))
        return out

    def win_prior_set(self, prior):
        if prior is None or prior.n == 0:
            self._call("win_prior_set", None)
        else:
            p = prior.c_struct()
            self._call("win_prior_set", C.byref(p))

    def download_state(self, w):
        s = w.c_state()
        self._call("download_state", C.byref(s))


def open_vilsolve(device=0, rank=0, world=1):
    return Backend(load_vilsolve(), "vil_", device, rank, world)

"""ctypes layer over include/vilvgicp.h (scan-to-scan voxelised GICP) + a synthetic scan pair.

`Vgicp(cdll, "vgicp_")` drives csrc/libvilsolve.so (HIP; needs a GPU, no CPU fallback);
`Vgicp(cdll, "orc_vgicp_")` drives oracle/liboracle.so -- tests / bench cpu_baseline leg only.
"""
import ctypes as C

import numpy as np

DIRECT1, DIRECT7, DIRECT27 = 1, 7, 27
LM, GN = 0, 1


class VgicpOptions(C.Structure):
    _fields_ = [("neighbor_mode", C.c_int32), ("optimizer", C.c_int32), ("max_iterations", C.c_int32), ("lm_max_iterations", C.c_int32),
                ("rotation_epsilon", C.c_double), ("transformation_epsilon", C.c_double), ("lm_init_lambda_factor", C.c_double)]


class VgicpSummary(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("converged", C.c_int32), ("n_correspondences", C.c_int32), ("lm_failed", C.c_int32),
                ("final_error", C.c_double), ("final_hessian", C.c_double * 36)]


class VgicpError(RuntimeError):
    pass


_dp, _fp = C.POINTER(C.c_double), C.POINTER(C.c_float)


class Vgicp:
    def __init__(self, cdll, prefix="vgicp_", device=0):
        self.lib, self.prefix = cdll, prefix
        self.ctx = C.c_void_p()
        st = self._f("create")(C.c_int32(device), C.byref(self.ctx))
        if st != 0:
            self.ctx = None
            raise VgicpError("%screate failed: status %d (no HIP device? there is no CPU fallback)" % (prefix, st))

    def _f(self, name):
        fc = self.__dict__.setdefault("_fcache", {})
        f = fc.get(name)
        if f is None:
            f = getattr(self.lib, self.prefix + name)
            f.restype = C.c_int
            fc[name] = f
        return f

    def close(self):
        if self.ctx is not None:
            f = getattr(self.lib, self.prefix + "destroy"); f.restype = None
            f(self.ctx); self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def default_options(self, **kw):
        o = VgicpOptions()
        f = getattr(self.lib, self.prefix + "default_options"); f.restype = None
        f(C.byref(o))
        for k, v in kw.items():
            if k not in dict(VgicpOptions._fields_):
                raise AttributeError("vgicp_options has no field %r" % k)
            setattr(o, k, v)
        return o

    def _chk(self, name, st):
        if st != 0:
            raise VgicpError("%s%s failed: status %d" % (self.prefix, name, st))

    def set_target(self, xyz, cov=None, resolution=0.5):
        """cov None: the library estimates the covariances itself (k = 20 neighbours, PLANE regularisation)."""
        xyz = np.ascontiguousarray(xyz, np.float32)
        cp = C.cast(None, _dp) if cov is None else np.ascontiguousarray(cov, np.float64).ctypes.data_as(_dp)
        self._chk("set_target", self._f("set_target")(self.ctx, C.c_int32(len(xyz)), xyz.ctypes.data_as(_fp), cp, C.c_double(resolution)))

    def set_source(self, xyz, cov=None):
        xyz = np.ascontiguousarray(xyz, np.float32)
        cp = C.cast(None, _dp) if cov is None else np.ascontiguousarray(cov, np.float64).ctypes.data_as(_dp)
        self._chk("set_source", self._f("set_source")(self.ctx, C.c_int32(len(xyz)), xyz.ctypes.data_as(_fp), cp))

    def covariances(self, xyz, k=20):
        xyz = np.ascontiguousarray(xyz, np.float32)
        out = np.zeros((len(xyz), 9))
        self._chk("covariances", self._f("covariances")(self.ctx, C.c_int32(len(xyz)), xyz.ctypes.data_as(_fp), C.c_int32(k), out.ctypes.data_as(_dp)))
        return out

    def linearize(self, T, mode=DIRECT1, jac=True):
        T = np.ascontiguousarray(T, np.float64)
        err, nc = C.c_double(), C.c_int32()
        H, b = np.zeros((6, 6)), np.zeros(6)
        vp = C.c_void_p                      # (plain addresses: a typed ctypes pointer per array costs 3 us, the call itself 20)
        self._chk("linearize", self._f("linearize")(self.ctx, vp(T.ctypes.data), C.c_int32(mode), C.byref(err), vp(H.ctypes.data) if jac else vp(None), vp(b.ctypes.data) if jac else vp(None), C.byref(nc)))
        return err.value, H, b, nc.value

    def compute_error(self, T):
        T = np.ascontiguousarray(T, np.float64)
        err = C.c_double()
        self._chk("compute_error", self._f("compute_error")(self.ctx, T.ctypes.data_as(_dp), C.byref(err)))
        return err.value

    def align(self, guess, opts=None):
        guess = np.ascontiguousarray(guess, np.float64)
        if opts is None:                       # the library only reads the options: one default record per wrapper
            opts = self.__dict__.get("_defopts")
            if opts is None:
                opts = self._defopts = self.default_options()
        T = np.empty((4, 4)); s = VgicpSummary()
        self._chk("align", self._f("align")(self.ctx, C.c_void_p(guess.ctypes.data), C.byref(opts), C.c_void_p(T.ctypes.data), C.byref(s)))
        return T, s


# ---- synthetic scan pair: a 16-ring spinning LiDAR in the 20 x 20 x 5 m room of synth.py, two poses -------------------------
def _rot(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]); Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]); Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def scan(pose_R, pose_t, rng, rings=16, az=900, noise=0.01):
    """Points (sensor frame, float32) + PLANE-regularised covariances R diag(1, 1, 1e-3) R^T (fast_gicp_impl.hpp:283-296) whose
    small axis is the wall normal seen from the sensor (perturbed by ~1 degree, like a kNN estimate would be)."""
    lo, hi = np.array([-10.0, -10.0, -1.5]), np.array([10.0, 10.0, 3.5])
    el = np.deg2rad(np.linspace(-15, 15, rings)); a = np.linspace(0, 2 * np.pi, az, endpoint=False)
    d = np.stack([np.outer(np.cos(el), np.cos(a)).ravel(), np.outer(np.cos(el), np.sin(a)).ravel(), np.outer(np.sin(el), np.ones_like(a)).ravel()], axis=1)
    dw = d @ pose_R.T
    with np.errstate(divide="ignore", invalid="ignore"):
        t_lo, t_hi = (lo - pose_t) / dw, (hi - pose_t) / dw
    tt = np.where(dw > 0, t_hi, t_lo)
    axis = np.argmin(tt, axis=1); r = tt[np.arange(len(tt)), axis]
    r = r + rng.normal(0, noise, len(r))
    pts = d * r[:, None]
    nrm_w = np.zeros_like(d); nrm_w[np.arange(len(d)), axis] = 1.0
    nrm = nrm_w @ pose_R + rng.normal(0, 0.02, d.shape)                  # sensor frame
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    cov = np.zeros((len(d), 3, 3))
    tmp = np.where(np.abs(nrm[:, :1]) < 0.9, np.array([[1.0, 0, 0]]), np.array([[0, 1.0, 0]]))
    u = np.cross(nrm, tmp); u /= np.linalg.norm(u, axis=1, keepdims=True); v = np.cross(nrm, u)
    for vec, val in ((u, 1.0), (v, 1.0), (nrm, 1e-3)):
        cov += val * vec[:, :, None] * vec[:, None, :]
    return pts.astype(np.float32), cov.reshape(-1, 9)


def make_pair(seed=0, rings=16, az=900, dt=(0.12, -0.05, 0.02), drot=(0.01, -0.008, 0.03)):
    """target scan at pose A, source scan at pose B; returns (tgt_xyz, tgt_cov, src_xyz, src_cov, T_true) with
    p_target = T_true p_source."""
    rng = np.random.default_rng(seed)
    RA, tA = _rot(0.01, -0.02, 0.3), np.array([1.0, -2.0, 0.2])
    Rd, td = _rot(*drot), np.array(dt)
    RB, tB = RA @ Rd, tA + RA @ td
    tx, tc = scan(RA, tA, rng, rings, az)
    sx, sc = scan(RB, tB, rng, rings, az)
    T = np.eye(4); T[:3, :3] = Rd; T[:3, 3] = td
    return tx, tc, sx, sc, T

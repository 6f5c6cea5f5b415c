"""ctypes mirror of include/vilsolve.h -- the C-ABI structs of the drop-in boundary.

Python is only the test/bench harness language here (the reference is C++; the host-side mirror a
maintainer links against is include/vilsolve.h + include/vilsolve_shim.hpp).
`Window` owns numpy arrays for one sliding window (what Estimator holds in estimator.h:67-168) and
hands out the POD structs.
"""
import ctypes as C

import numpy as np

VIL_IMU_CONST, VIL_VIS_CONST, VIL_EDGE_CONST, VIL_PLANE_CONST, VIL_ICP_CONST, VIL_LPS_CONST = 287, 14, 9, 7, 10, 7
VIL_MAX_TRACE = 64
(FACTOR_IMU, FACTOR_VISUAL, FACTOR_PRIOR, FACTOR_ICP, FACTOR_LPS, FACTOR_EDGE, FACTOR_PLANE) = range(7)
NR = {FACTOR_IMU: 15, FACTOR_VISUAL: 2, FACTOR_ICP: 3, FACTOR_LPS: 3, FACTOR_EDGE: 3, FACTOR_PLANE: 1}
NJ = {FACTOR_IMU: 480, FACTOR_VISUAL: 46, FACTOR_ICP: 84, FACTOR_LPS: 42, FACTOR_EDGE: 21, FACTOR_PLANE: 7}
BLK_POSE, BLK_SPEEDBIAS, BLK_EX, BLK_TD = range(4)
LOSS_NONE, LOSS_CAUCHY, LOSS_HUBER = range(3)
MARGIN_OLD, MARGIN_SECOND_NEW = 0, 1
TERM_NAMES = ["none", "function_tolerance", "gradient_tolerance", "parameter_tolerance", "max_iterations", "max_time", "failure"]

# pointer fields of the C structs: declared void* here (same layout) so that a field takes the array's address as a plain integer --
# building a typed ctypes pointer per table costs 3 us, and the per-image loop of the replay fills ~40 of them
_dp = C.c_void_p
_ip = C.c_void_p
_bp = C.c_void_p


class VilState(C.Structure):
    _fields_ = [("K", C.c_int32), ("L", C.c_int32), ("pose", _dp), ("speedbias", _dp), ("ex_pose", _dp), ("td", _dp), ("inv_depth", _dp)]


class VilPrior(C.Structure):
    _fields_ = [("n", C.c_int32), ("nblk", C.c_int32), ("blk_kind", _ip), ("blk_index", _ip), ("blk_col", _ip), ("x0", _dp), ("J0", _dp), ("r0", _dp)]


class VilProblem(C.Structure):
    _fields_ = [
        ("K", C.c_int32), ("L", C.c_int32),
        ("pose_const", _bp), ("sb_const", _bp), ("lm_const", _bp),
        ("ex_const", C.c_int32), ("td_const", C.c_int32), ("use_td", C.c_int32),
        ("n_imu", C.c_int32), ("imu_i", _ip), ("imu_j", _ip), ("imu_const", _dp),
        ("n_vis", C.c_int32), ("vis_i", _ip), ("vis_j", _ip), ("vis_l", _ip), ("vis_const", _dp),
        ("prior", VilPrior),
        ("n_icp", C.c_int32), ("icp_ids", _ip), ("icp_const", _dp),
        ("n_lps", C.c_int32), ("lps_ids", _ip), ("lps_const", _dp),
        ("n_edge", C.c_int32), ("edge_pose", _ip), ("edge_const", _dp),
        ("n_plane", C.c_int32), ("plane_pose", _ip), ("plane_const", _dp),
        ("q_lb", C.c_double * 4), ("t_lb", C.c_double * 3),
        ("G", C.c_double * 3), ("sqrt_info_px", C.c_double), ("tr_over_row", C.c_double),
    ]


class VilOptions(C.Structure):
    _fields_ = [
        ("max_iterations", C.c_int32), ("max_time_s", C.c_double),
        ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double), ("parameter_tolerance", C.c_double),
        ("initial_radius", C.c_double), ("max_radius", C.c_double), ("min_relative_decrease", C.c_double),
        ("min_mu", C.c_double), ("max_mu", C.c_double), ("jacobi_scaling", C.c_int32),
        ("visual_loss", C.c_int32), ("visual_loss_scale", C.c_double),
        ("lidar_loss", C.c_int32), ("lidar_loss_scale", C.c_double),
        ("rel_loss", C.c_int32), ("rel_loss_scale", C.c_double),
        ("autodiff_quirk", C.c_int32), ("precision", C.c_int32),
    ]


class VilSummary(C.Structure):
    _fields_ = [
        ("iterations", C.c_int32), ("successful_steps", C.c_int32), ("termination", C.c_int32),
        ("initial_cost", C.c_double), ("final_cost", C.c_double),
        ("t_prepare_ms", C.c_double), ("t_solve_ms", C.c_double), ("t_readback_ms", C.c_double),
        ("cost_trace", C.c_double * VIL_MAX_TRACE), ("radius_trace", C.c_double * VIL_MAX_TRACE),
    ]


class VilMargSpec(C.Structure):
    _fields_ = [("flag", C.c_int32), ("icp_marg", C.c_int32), ("lps_marg", C.c_int32), ("threads", C.c_int32)]


class VilPriorOut(C.Structure):
    _fields_ = [("n", C.c_int32), ("nblk", C.c_int32), ("m", C.c_int32), ("blk_kind", _ip), ("blk_index", _ip), ("blk_col", _ip),
                ("x0", _dp), ("J0", _dp), ("r0", _dp), ("A", _dp), ("b", _dp)]


VIL_WIN_OBS, VIL_WIN_MAXBLK = 8, 24


class VilWinCfg(C.Structure):
    _fields_ = [("K", C.c_int32), ("max_tracks", C.c_int32), ("max_samples", C.c_int32), ("use_td", C.c_int32), ("noise", C.c_double * 4),
                ("G", C.c_double * 3), ("sqrt_info_px", C.c_double), ("tr_over_row", C.c_double), ("q_lb", C.c_double * 4), ("t_lb", C.c_double * 3)]


class VilWinFrame(C.Structure):
    _fields_ = [("n_samples", C.c_int32), ("dt", _dp), ("acc", _dp), ("gyr", _dp),
                ("acc0", C.c_double * 3), ("gyr0", C.c_double * 3), ("lin_ba", C.c_double * 3), ("lin_bg", C.c_double * 3),
                ("n_obs", C.c_int32), ("obs_track", _ip), ("obs", _dp),
                ("n_plane", C.c_int32), ("plane_const", _dp), ("n_edge", C.c_int32), ("edge_const", _dp)]


class VilWinProblem(C.Structure):
    _fields_ = [("L", C.c_int32), ("lm_track", _ip), ("lm_start", _ip), ("lm_nobs", _ip), ("lm_const", _bp),
                ("pose_const", _bp), ("sb_const", _bp), ("ex_const", C.c_int32), ("td_const", C.c_int32),
                ("n_icp", C.c_int32), ("icp_ids", _ip), ("icp_const", _dp), ("n_lps", C.c_int32), ("lps_ids", _ip), ("lps_const", _dp)]


class VilWinPriorInfo(C.Structure):
    _fields_ = [("n", C.c_int32), ("nblk", C.c_int32), ("m", C.c_int32), ("blk_kind", C.c_int32 * VIL_WIN_MAXBLK), ("blk_index", C.c_int32 * VIL_WIN_MAXBLK), ("blk_col", C.c_int32 * VIL_WIN_MAXBLK)]


class VilDeviceCfg(C.Structure):
    _fields_ = [("device", C.c_int32), ("rank", C.c_int32), ("world", C.c_int32), ("reserved", C.c_int32)]


def _d(a):
    # (the buffer address without building a ctypes proxy object: ndarray.ctypes costs ~2 us per array, and a window hands over twenty of them per call)
    return a.__array_interface__["data"][0] if a is not None and a.size else None


_i = _d
_b = _d


def f64(a, shape=None):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    return a.reshape(shape) if shape is not None else a


def i32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


def default_options(**kw):
    """ceres::Solver::Options as used at estimator.cpp:1400-1411 (time cap disabled for reproducibility)."""
    o = VilOptions()
    o.max_iterations = 30
    o.max_time_s = 0.0
    o.function_tolerance, o.gradient_tolerance, o.parameter_tolerance = 1e-6, 1e-10, 1e-8
    o.initial_radius, o.max_radius, o.min_relative_decrease = 1e4, 1e16, 1e-3
    o.min_mu, o.max_mu, o.jacobi_scaling = 1e-8, 1.0, 1
    o.visual_loss, o.visual_loss_scale = LOSS_CAUCHY, 1.0
    o.lidar_loss, o.lidar_loss_scale = LOSS_HUBER, 0.1
    o.rel_loss, o.rel_loss_scale = LOSS_CAUCHY, 1.0
    o.autodiff_quirk, o.precision = 1, 0
    for k, v in kw.items():
        if k not in dict(VilOptions._fields_):
            raise AttributeError("vil_options has no field %r" % k)
        setattr(o, k, v)
    return o


class Prior:
    """MarginalizationInfo's linearised output (marginalization_factor.h:60-75)."""

    def __init__(self, n=0, blk_kind=(), blk_index=(), blk_col=(), x0=(), J0=None, r0=()):
        self.n = int(n)
        self.blk_kind, self.blk_index, self.blk_col = i32(blk_kind), i32(blk_index), i32(blk_col)
        self.x0 = f64(x0)
        self.J0 = f64(J0 if J0 is not None else np.zeros((0, 0)))  # stored COLUMN-major flat: J0[k*n+i] = J(i,k)
        self.r0 = f64(r0)

    def J_matrix(self):
        return self.J0.reshape(self.n, self.n).T  # J(i,k)

    def c_struct(self):
        p = VilPrior()
        p.n, p.nblk = self.n, len(self.blk_kind)
        p.blk_kind, p.blk_index, p.blk_col = _i(self.blk_kind), _i(self.blk_index), _i(self.blk_col)
        p.x0, p.J0, p.r0 = _d(self.x0), _d(self.J0), _d(self.r0)
        return p


class Window:
    """One sliding window: state + factor tables (numpy, caller-owned) -> vil_problem / vil_state."""

    def __init__(self, K, L):
        self.K, self.L = int(K), int(L)
        self.pose = np.zeros((K, 7)); self.pose[:, 6] = 1.0
        self.speedbias = np.zeros((K, 9))
        self.ex_pose = np.array([0, 0, 0, 0, 0, 0, 1.0])
        self.td = np.zeros(1)
        self.inv_depth = np.full(L, 0.2)
        self.pose_const = np.zeros(K, np.uint8)
        self.sb_const = np.zeros(K, np.uint8)
        self.lm_const = np.zeros(L, np.uint8)
        self.ex_const, self.td_const, self.use_td = 0, 0, 1
        self.imu_i = i32([]); self.imu_j = i32([]); self.imu_const = np.zeros((0, VIL_IMU_CONST))
        self.vis_i = i32([]); self.vis_j = i32([]); self.vis_l = i32([]); self.vis_const = np.zeros((0, VIL_VIS_CONST))
        self.prior = Prior()
        self.icp_ids = np.zeros((0, 4), np.int32); self.icp_const = np.zeros((0, VIL_ICP_CONST))
        self.lps_ids = np.zeros((0, 2), np.int32); self.lps_const = np.zeros((0, VIL_LPS_CONST))
        self.edge_pose = i32([]); self.edge_const = np.zeros((0, VIL_EDGE_CONST))
        self.plane_pose = i32([]); self.plane_const = np.zeros((0, VIL_PLANE_CONST))
        self.q_lb = np.array([0, 0, 0, 1.0]); self.t_lb = np.zeros(3)
        self.G = np.array([0, 0, 9.795])
        self.sqrt_info_px = 230.0
        self.tr_over_row = 0.0
        self.truth = None  # optional dict with ground-truth arrays (generator only)
        self.lidar_resident = False  # True: the point factors are the context's frame slabs (vil_lidar_push), n_plane = n_edge = VIL_LIDAR_RESIDENT

    # -- normalise dtypes/contiguity so the pointers stay valid for the struct's lifetime
    def _fix(self):
        def ok(a, dt):      # already what the struct needs: nothing to copy (the per-image loop of the replay calls this twice per image)
            return isinstance(a, np.ndarray) and a.dtype == dt and a.flags.c_contiguous
        for name in ("pose", "speedbias", "ex_pose", "td", "inv_depth", "imu_const", "vis_const", "icp_const", "lps_const",
                     "edge_const", "plane_const", "q_lb", "t_lb", "G"):
            a = getattr(self, name)
            if not ok(a, np.float64):
                setattr(self, name, f64(a))
        for name in ("imu_i", "imu_j", "vis_i", "vis_j", "vis_l", "icp_ids", "lps_ids", "edge_pose", "plane_pose"):
            a = getattr(self, name)
            if not ok(a, np.int32):
                setattr(self, name, i32(a))
        for name in ("pose_const", "sb_const", "lm_const"):
            a = getattr(self, name)
            if not ok(a, np.uint8):
                setattr(self, name, np.ascontiguousarray(np.asarray(a, dtype=np.uint8)))

    def c_state(self):
        self._fix()
        s = VilState()
        s.K, s.L = self.K, self.L
        s.pose, s.speedbias, s.ex_pose, s.td, s.inv_depth = _d(self.pose), _d(self.speedbias), _d(self.ex_pose), _d(self.td), _d(self.inv_depth)
        return s

    def c_problem(self):
        self._fix()
        p = VilProblem()
        p.K, p.L = self.K, self.L
        p.pose_const, p.sb_const, p.lm_const = _b(self.pose_const), _b(self.sb_const), _b(self.lm_const)
        p.ex_const, p.td_const, p.use_td = int(self.ex_const), int(self.td_const), int(self.use_td)
        p.n_imu = len(self.imu_i); p.imu_i, p.imu_j, p.imu_const = _i(self.imu_i), _i(self.imu_j), _d(self.imu_const)
        p.n_vis = len(self.vis_i); p.vis_i, p.vis_j, p.vis_l, p.vis_const = _i(self.vis_i), _i(self.vis_j), _i(self.vis_l), _d(self.vis_const)
        p.prior = self.prior.c_struct()
        p.n_icp = len(self.icp_ids); p.icp_ids, p.icp_const = _i(self.icp_ids), _d(self.icp_const)
        p.n_lps = len(self.lps_ids); p.lps_ids, p.lps_const = _i(self.lps_ids), _d(self.lps_const)
        p.n_edge = len(self.edge_pose); p.edge_pose, p.edge_const = _i(self.edge_pose), _d(self.edge_const)
        p.n_plane = len(self.plane_pose); p.plane_pose, p.plane_const = _i(self.plane_pose), _d(self.plane_const)
        if getattr(self, "lidar_resident", False):
            p.n_plane = p.n_edge = -1
            p.plane_pose = p.edge_pose = None; p.plane_const = p.edge_const = None
        for k in range(4):
            p.q_lb[k] = self.q_lb[k]
        for k in range(3):
            p.t_lb[k] = self.t_lb[k]; p.G[k] = self.G[k]
        p.sqrt_info_px, p.tr_over_row = float(self.sqrt_info_px), float(self.tr_over_row)
        return p

    @property
    def D(self):
        return 15 * self.K + 7

    def state_copy(self):
        return dict(pose=self.pose.copy(), speedbias=self.speedbias.copy(), ex_pose=self.ex_pose.copy(), td=self.td.copy(), inv_depth=self.inv_depth.copy())

    def set_state(self, s):
        self.pose, self.speedbias, self.ex_pose, self.td, self.inv_depth = (f64(s[k]).copy() for k in ("pose", "speedbias", "ex_pose", "td", "inv_depth"))

    _ARRAYS = ("pose", "speedbias", "ex_pose", "td", "inv_depth", "pose_const", "sb_const", "lm_const", "imu_i", "imu_j", "imu_const",
               "vis_i", "vis_j", "vis_l", "vis_const", "icp_ids", "icp_const", "lps_ids", "lps_const", "edge_pose", "edge_const",
               "plane_pose", "plane_const", "q_lb", "t_lb", "G")

    def to_dict(self):
        """Plain-data image of the window (for golden fixtures)."""
        self._fix()
        d = {k: getattr(self, k) for k in self._ARRAYS}
        d["scalars"] = np.array([self.K, self.L, self.ex_const, self.td_const, self.use_td, self.sqrt_info_px, self.tr_over_row], dtype=np.float64)
        pr = self.prior
        d.update(prior_n=np.array([pr.n]), prior_kind=pr.blk_kind, prior_index=pr.blk_index, prior_col=pr.blk_col, prior_x0=pr.x0, prior_J0=pr.J0, prior_r0=pr.r0)
        return d

    @classmethod
    def from_dict(cls, d):
        sc = d["scalars"]
        w = cls(int(sc[0]), int(sc[1]))
        for k in cls._ARRAYS:
            setattr(w, k, np.array(d[k]))
        w.ex_const, w.td_const, w.use_td, w.sqrt_info_px, w.tr_over_row = int(sc[2]), int(sc[3]), int(sc[4]), float(sc[5]), float(sc[6])
        w.prior = Prior(int(d["prior_n"][0]), d["prior_kind"], d["prior_index"], d["prior_col"], d["prior_x0"], d["prior_J0"], d["prior_r0"])
        w._fix()
        return w

    def nfactors(self, cls):
        return {FACTOR_IMU: len(self.imu_i), FACTOR_VISUAL: len(self.vis_i), FACTOR_ICP: len(self.icp_ids), FACTOR_LPS: len(self.lps_ids),
                FACTOR_EDGE: len(self.edge_pose), FACTOR_PLANE: len(self.plane_pose), FACTOR_PRIOR: 1 if self.prior.n else 0}[cls]

    def eval_sizes(self, cls):
        """(#residual doubles, #jacobian doubles) vil_eval_factors writes for a class."""
        if cls == FACTOR_PRIOR:
            n = self.prior.n
            gs = sum({BLK_POSE: 7, BLK_SPEEDBIAS: 9, BLK_EX: 7, BLK_TD: 1}[int(k)] for k in self.prior.blk_kind)
            return n, n * gs
        return self.nfactors(cls) * NR[cls], self.nfactors(cls) * NJ[cls]


class PriorOut:
    """Caller-provided storage for vil_marginalize / orc_marginalize."""

    def __init__(self, K):
        self.n_max = 6 * K + 16
        self.nblk_max = K + 4
        self.x0_max = 7 * K + 9 + 7 + 1 + 16
        self.blk_kind = np.zeros(self.nblk_max, np.int32)
        self.blk_index = np.zeros(self.nblk_max, np.int32)
        self.blk_col = np.zeros(self.nblk_max, np.int32)
        self.x0 = np.zeros(self.x0_max)
        self.J0 = np.zeros(self.n_max * self.n_max)
        self.r0 = np.zeros(self.n_max)
        self.A = np.zeros(self.n_max * self.n_max)
        self.b = np.zeros(self.n_max)
        self.c = VilPriorOut()
        self.c.blk_kind, self.c.blk_index, self.c.blk_col = _i(self.blk_kind), _i(self.blk_index), _i(self.blk_col)
        self.c.x0, self.c.J0, self.c.r0, self.c.A, self.c.b = _d(self.x0), _d(self.J0), _d(self.r0), _d(self.A), _d(self.b)

    def to_prior(self):
        n, nb = self.c.n, self.c.nblk
        if n < 0:
            return None
        kinds = self.blk_kind[:nb].copy()
        gs = sum({BLK_POSE: 7, BLK_SPEEDBIAS: 9, BLK_EX: 7, BLK_TD: 1}[int(k)] for k in kinds)
        return Prior(n, kinds, self.blk_index[:nb].copy(), self.blk_col[:nb].copy(), self.x0[:gs].copy(), self.J0[: n * n].copy(), self.r0[:n].copy())

    def A_matrix(self):
        n = self.c.n
        return self.A[: n * n].reshape(n, n).copy()

    def b_vector(self):
        return self.b[: self.c.n].copy()

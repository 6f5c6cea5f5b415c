"""ctypes layer over include/vilmap.h (LiDAR scan-to-map registration) + a synthetic local map / scan.

`MapReg(cdll, "vmap_")` drives csrc/libvilsolve.so (HIP; needs a GPU, no CPU fallback); `MapReg(cdll, "orc_vmap_")` drives
oracle/liboracle.so -- tests / bench cpu_baseline leg only.
"""
import ctypes as C

import numpy as np

from . import abi
from .vgicp import _rot


class VmapSummary(C.Structure):
    _fields_ = [("rounds", C.c_int32), ("n_edge", C.c_int32), ("n_plane", C.c_int32), ("iterations", C.c_int32),
                ("initial_cost", C.c_double), ("final_cost", C.c_double),
                ("t_associate_ms", C.c_double), ("t_prepare_ms", C.c_double), ("t_solve_ms", C.c_double)]


class MapRegError(RuntimeError):
    pass


_dp, _fp = C.POINTER(C.c_double), C.POINTER(C.c_float)


class MapReg:
    def __init__(self, cdll, prefix="vmap_", device=0):
        self.lib, self.prefix = cdll, prefix
        self.ctx = C.c_void_p()
        st = self._f("create")(C.c_int32(device), C.byref(self.ctx))
        if st != 0:
            self.ctx = None
            raise MapRegError("%screate failed: status %d (no HIP device? there is no CPU fallback)" % (prefix, st))

    def _f(self, name):
        f = getattr(self.lib, self.prefix + name)
        f.restype = C.c_int
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

    def _chk(self, name, st):
        if st != 0:
            raise MapRegError("%s%s failed: status %d" % (self.prefix, name, st))

    def set_map(self, corner, surf):
        corner = np.ascontiguousarray(corner, np.float32).reshape(-1, 4); surf = np.ascontiguousarray(surf, np.float32).reshape(-1, 4)
        self._chk("set_map", self._f("set_map")(self.ctx, C.c_int32(len(corner)), corner.ctypes.data_as(_fp), C.c_int32(len(surf)), surf.ctypes.data_as(_fp)))

    def associate(self, corner, surf, q, t):
        corner = np.ascontiguousarray(corner, np.float32).reshape(-1, 4); surf = np.ascontiguousarray(surf, np.float32).reshape(-1, 4)
        q = np.ascontiguousarray(q, np.float64); t = np.ascontiguousarray(t, np.float64)
        ne, npl = C.c_int32(), C.c_int32()
        edge = np.zeros((max(1, len(corner)), 9)); plane = np.zeros((max(1, len(surf)), 7))
        self._chk("associate", self._f("associate")(self.ctx, C.c_int32(len(corner)), corner.ctypes.data_as(_fp), C.c_int32(len(surf)), surf.ctypes.data_as(_fp),
                                                    q.ctypes.data_as(_dp), t.ctypes.data_as(_dp), C.byref(ne), edge.ctypes.data_as(_dp), C.byref(npl), plane.ctypes.data_as(_dp)))
        return edge[:ne.value].copy(), plane[:npl.value].copy()

    def align(self, solver_ctx, corner, surf, q, t, opts=None):
        """solver_ctx: the vil_ctx of a lib.Backend (None for the oracle).  Returns (q, t, summary)."""
        corner = np.ascontiguousarray(corner, np.float32).reshape(-1, 4); surf = np.ascontiguousarray(surf, np.float32).reshape(-1, 4)
        q = np.array(q, np.float64); t = np.array(t, np.float64)
        opts = opts or abi.default_options(max_iterations=4)          # localMapping.cpp:772
        s = VmapSummary()
        self._chk("align", self._f("align")(self.ctx, solver_ctx, C.c_int32(len(corner)), corner.ctypes.data_as(_fp), C.c_int32(len(surf)), surf.ctypes.data_as(_fp),
                                            q.ctypes.data_as(_dp), t.ctypes.data_as(_dp), C.byref(opts), C.byref(s)))
        return q, t, s


# ---- synthetic local map of the 20 x 20 x 5 m room: surf points on the walls / floor / ceiling, corner points on its 12 edges --
def make_map(seed=0, n_surf=20000, n_corner=3000):
    rng = np.random.default_rng(seed)
    lo, hi = np.array([-10.0, -10.0, -1.5]), np.array([10.0, 10.0, 3.5])
    axis = rng.integers(0, 3, n_surf); side = rng.integers(0, 2, n_surf)
    P = rng.uniform(lo, hi, (n_surf, 3)); P[np.arange(n_surf), axis] = np.where(side == 1, hi[axis], lo[axis])
    P += rng.normal(0, 0.01, P.shape)
    surf = np.concatenate([P, (10.0 * (2 * axis + side) + rng.uniform(0, 5, n_surf))[:, None]], axis=1)        # intensity: one band per wall
    ax = rng.integers(0, 3, n_corner); Cp = rng.uniform(lo, hi, (n_corner, 3))
    code = np.zeros(n_corner)
    for a in range(3):
        m = ax != a; sel = rng.integers(0, 2, n_corner)
        Cp[m, a] = np.where(sel[m] == 1, hi[a], lo[a]); code += np.where(m, sel * (2 ** a), 0)
    Cp += rng.normal(0, 0.01, Cp.shape)
    corner = np.concatenate([Cp, (10.0 * ax + code + rng.uniform(0, 0.5, n_corner))[:, None]], axis=1)
    return corner.astype(np.float32), surf.astype(np.float32)


def make_scan(corner_map, surf_map, R, t, seed=1, n_surf=4000, n_corner=600, max_range=12.0):
    """A scan taken at world pose (R, t): map points within range, in the sensor frame, with 1 cm noise (xyz) and jittered intensity."""
    rng = np.random.default_rng(seed)
    out = []
    for M, n in ((corner_map, n_corner), (surf_map, n_surf)):
        X = M[:, :3].astype(np.float64)
        near = np.where(np.linalg.norm(X - t, axis=1) < max_range)[0]
        pick = rng.choice(near, min(n, len(near)), replace=False)
        Xs = (X[pick] + rng.normal(0, 0.05, (len(pick), 3)) - t) @ R + rng.normal(0, 0.01, (len(pick), 3))    # not the map points themselves
        out.append(np.concatenate([Xs, (M[pick, 3] + rng.normal(0, 0.3, len(pick)))[:, None]], axis=1).astype(np.float32))
    return out[0], out[1]


def quat_from_R(R):
    from .synth import R_to_quat
    return R_to_quat(R)

"""Host logic of the sweep's visual role (vil_visual_plan, include/vilsolve.h; csrc/vilsolve.hip: plan_visual) -- no GPU: the landmarks are sorted by
(first frame, last frame), the factor tables stored in that order and cut into chunks, one per workgroup, whose record is the upper 16 x 16 tiles of the
chunk's own frame window.  Invariants of the plan on the BASELINE configs and on ragged / degenerate windows."""
import ctypes as C

import numpy as np
import pytest

from mvil_fusion_amd import abi, lib, synth

VIS_LM, VIS_MF, VIS_GM = 16, 64, 16384


class PlanInfo(C.Structure):
    _fields_ = [("n_chunks", C.c_int32), ("max_tiles", C.c_int32), ("tiles_per_wave", C.c_int32), ("cost_cap", C.c_int32),
                ("record_bytes", C.c_int64), ("dense_record_bytes", C.c_int64), ("lds_bytes", C.c_int64)]


def plan(w):
    so = C.CDLL(lib.LIB_PATH)
    p = w.c_problem()
    info = PlanInfo()
    cap = max(1, len(w.vis_i))
    fa, sp, nf, nl = (np.zeros(cap, np.int32) for _ in range(4))
    pos = np.zeros(max(1, len(w.vis_i)), np.int32)
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    st = so.vil_visual_plan(C.byref(p), C.byref(info), C.c_int32(cap), ip(fa), ip(sp), ip(nf), ip(nl), ip(pos))
    assert st == 0, st
    n = info.n_chunks
    return info, fa[:n], sp[:n], nf[:n], nl[:n], pos[:len(w.vis_i)]


def tiles(span):
    return (6 * span + 8 + 15) // 16


def check(w):
    info, fa, sp, nf, nl, pos = plan(w)
    F = len(w.vis_i)
    assert nf.sum() == F and sorted(pos.tolist()) == list(range(F))                    # every factor in exactly one chunk, the positions a permutation
    assert (nl <= VIS_LM).all() and (nf <= VIS_MF).all() and (nl >= 1).all()
    assert (np.diff(fa) >= 0).all()                                                    # chunks sorted by first frame (the gather's prefix bound relies on it)
    # a landmark's factors are consecutive sorted positions, and the chunk that holds them covers all their frames
    start = np.concatenate([[0], np.cumsum(nf)])
    chunk_of = np.searchsorted(start, pos, side="right") - 1
    for l in np.unique(w.vis_l):
        m = np.where(w.vis_l == l)[0]
        assert np.array_equal(np.sort(pos[m]), np.arange(pos[m].min(), pos[m].min() + len(m)))
        cs = np.unique(chunk_of[m]); assert len(cs) == 1
        c = cs[0]
        lo, hi = min(w.vis_i[m].min(), w.vis_j[m].min()), max(w.vis_i[m].max(), w.vis_j[m].max())
        assert fa[c] <= lo and hi < fa[c] + sp[c]
    # LDS: the operand rows of the largest chunk fit, record bytes as the device lays them out
    T = np.array([tiles(s) for s in sp])
    rs = 16 * ((T + 1) | 1)
    gm = (((2 * nf + 15) // 16) * 16 + 16) * rs + 16
    assert (gm <= VIS_GM).all() and info.lds_bytes <= 160 * 1024
    assert info.max_tiles == T.max() and info.tiles_per_wave == (2 if (T.max() * (T.max() + 1) // 2 + 7) // 8 <= 2 else 5)
    assert info.record_bytes == 8 * int((T * (T + 1) // 2 * 256 + 32 * T + 16).sum())
    return info, fa, sp, nf, nl


@pytest.mark.parametrize("cid", [1, 2, 3, 4])
def test_plan_of_the_baseline_configs(cid):
    w = synth.make_config(cid)
    info, fa, sp, nf, nl = check(w)
    if cid == 2:
        assert info.tiles_per_wave == 2 and info.n_chunks <= 128 and info.record_bytes < info.dense_record_bytes
    if cid == 4:                                                                        # K = 20: the numbers DESIGN.md section 0c quotes
        assert w.K == 20 and info.tiles_per_wave == 5 and info.max_tiles == 8
        assert 200 <= info.n_chunks <= 256
        assert 8.0e6 < info.record_bytes < 9.5e6 and info.dense_record_bytes > 1.5e7     # 8.8 MB of window records against 16 MB of packed triangles


def test_plan_of_ragged_and_degenerate_windows():
    w = synth.make_config(2, L=150, n_plane=0, n_edge=0)
    keep = (w.vis_l != 5) & (w.vis_l != 77)                                             # landmarks without factors are in no chunk
    w.vis_i, w.vis_j, w.vis_l, w.vis_const = w.vis_i[keep], w.vis_j[keep], w.vis_l[keep], w.vis_const[keep]
    check(w)
    w = synth.make_config(2, L=150, n_plane=0, n_edge=0)
    w.vis_i, w.vis_j, w.vis_l, w.vis_const = w.vis_i[:0], w.vis_j[:0], w.vis_l[:0], w.vis_const[:0]
    info, *_ = plan(w)
    assert info.n_chunks == 0 and info.record_bytes == 0
    w = synth.make_config(2, K=4, L=40, n_plane=0, n_edge=0)                             # the smallest window the synthetic tracks allow
    info, fa, sp, nf, nl = check(w)
    assert info.max_tiles <= 2

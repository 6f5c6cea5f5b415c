"""Error behaviour of the widened rows' C-ABIs on a GPU box: invalid arguments are refused with a status (no crash, no
partial output), calls in the wrong order are refused, and the context stays usable afterwards."""
import ctypes as C

import numpy as np
import pytest

from mvil_fusion_amd import abi, lib, mapreg, preint, vgicp

pytestmark = pytest.mark.gpu
_dp, _fp, _ip = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int32)


def test_preintegration_argument_checks():
    so = lib.load_vilsolve()
    p = preint.Preint(so, "vpre_")
    s = preint.make_stream(n_intervals=2, samples=(5, 5), seed=0)
    f = so.vpre_integrate; f.restype = C.c_int
    a = [np.ascontiguousarray(x, np.float64) for x in s[1:]] + [preint.NOISE.copy()]
    out = np.full((2, 287), -7.0)
    def call(start, out_ptr=out.ctypes.data_as(_dp), n=2):
        st = np.ascontiguousarray(start, np.int32)
        return f(p.ctx, C.c_int32(n), st.ctypes.data_as(_ip), *[x.ctypes.data_as(_dp) for x in a], out_ptr, None)
    assert call([1, 5, 10]) == -1 and call([0, 6, 4]) == -1 and call([0, 5, 10], out_ptr=None) == -1 and call([0, 5, 10], n=-1) == -1
    assert np.all(out == -7.0)                                              # nothing was written by the refused calls
    assert call([0, 5, 10]) == 0 and np.all(out[:, 16] > 0)                 # and the context still works
    assert f(None, C.c_int32(2), None, *[None] * 8, None, None) == -1
    p.close()


def test_mapreg_argument_checks(hip):
    so = lib.load_vilsolve()
    m = mapreg.MapReg(so, "vmap_")
    cm, sm = mapreg.make_map(seed=1, n_surf=2000, n_corner=400)
    sc, ss = cm[:50].copy(), sm[:200].copy()
    q, t = np.array([0, 0, 0, 1.0]), np.zeros(3)
    fa = so.vmap_associate; fa.restype = C.c_int
    ne, npl = C.c_int32(), C.c_int32()
    e9, p7 = np.zeros((50, 9)), np.zeros((200, 7))
    args = lambda qq: (m.ctx, C.c_int32(50), sc.ctypes.data_as(_fp), C.c_int32(200), ss.ctypes.data_as(_fp), qq, t.ctypes.data_as(_dp), C.byref(ne), e9.ctypes.data_as(_dp), C.byref(npl), p7.ctypes.data_as(_dp))
    assert fa(*args(None)) == -1                                            # no pose
    assert fa(*args(q.ctypes.data_as(_dp))) == 0 and ne.value == 0 and npl.value == 0      # no map yet: no factors, not an error
    assert so.vmap_set_map(m.ctx, C.c_int32(-1), None, C.c_int32(0), None) == -1
    m.set_map(cm, sm)
    fal = so.vmap_align; fal.restype = C.c_int
    s = mapreg.VmapSummary(); o = abi.default_options(max_iterations=4)
    qq, tt = q.copy(), t.copy()
    assert fal(m.ctx, None, C.c_int32(50), sc.ctypes.data_as(_fp), C.c_int32(200), ss.ctypes.data_as(_fp), qq.ctypes.data_as(_dp), tt.ctypes.data_as(_dp), C.byref(o), C.byref(s)) == -1   # no solver context
    assert np.array_equal(qq, q) and np.array_equal(tt, t)
    q2, t2, s2 = m.align(hip.ctx, sc, ss, q, t)                              # still usable
    assert s2.rounds == 2
    m.close()


def test_vgicp_call_order_and_arguments():
    so = lib.load_vilsolve()
    g = vgicp.Vgicp(so, "vgicp_")
    with pytest.raises(vgicp.VgicpError):
        g.linearize(np.eye(4))                                              # no target / source yet
    tx, tc, sx, sc, _ = vgicp.make_pair(seed=2, rings=4, az=120)
    g.set_target(tx, tc, 0.5)
    with pytest.raises(vgicp.VgicpError):
        g.linearize(np.eye(4))                                              # still no source
    g.set_source(sx, sc)
    with pytest.raises(vgicp.VgicpError):
        g.compute_error(np.eye(4))                                          # compute_error reuses the correspondences of a linearisation
    e, H, b, n = g.linearize(np.eye(4))
    assert n > 0 and abs(g.compute_error(np.eye(4)) - e) <= 1e-12 * e
    Tn = np.eye(4); Tn[0, 3] = np.nan
    with pytest.raises(vgicp.VgicpError):
        g.linearize(Tn)                                                     # non-finite transform: refused, not propagated
    g.close()

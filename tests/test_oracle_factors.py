"""Pins the CPU oracle's factor arithmetic (no GPU).

The reference has no tests; its only in-tree self-check is the finite-difference recipe of
ProjectionFactor::check / ProjectionTdFactor::check (projection_factor.cpp:123-225,
projection_td_factor.cpp:143-259): forward difference, P += d, Q = Q * deltaQ(d).  The same recipe
(central differences for accuracy) is applied to every analytic factor here, plus independent numpy
re-derivations of residuals.
"""
import numpy as np
import pytest

from mvil_fusion_amd import abi, synth


def plus_pose(p7, d6):
    out = p7.copy()
    out[:3] += d6[:3]
    q = synth.qmul(p7[3:], np.array([d6[3] / 2, d6[4] / 2, d6[5] / 2, 1.0]))
    out[3:] = q / np.linalg.norm(q)
    return out


def perturbed(w, kind, idx, col, eps):
    """Copy of window w with the LOCAL coordinate `col` of block (kind, idx) moved by eps."""
    w2 = synth.make_config.__globals__["Window"](w.K, w.L)
    w2.__dict__.update({k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in w.__dict__.items()})
    if kind == "pose":
        d = np.zeros(6); d[col] = eps
        w2.pose[idx] = plus_pose(w.pose[idx], d)
    elif kind == "sb":
        w2.speedbias[idx, col] += eps
    elif kind == "ex":
        d = np.zeros(6); d[col] = eps
        w2.ex_pose = plus_pose(w.ex_pose, d)
    elif kind == "td":
        w2.td[0] += eps
    elif kind == "lam":
        w2.inv_depth[idx] += eps
    return w2


def fd_column(oracle, w, cls, kind, idx, col, eps=1e-6):
    rp, _ = oracle.eval_factors(perturbed(w, kind, idx, col, eps), cls, jac=False)
    rm, _ = oracle.eval_factors(perturbed(w, kind, idx, col, -eps), cls, jac=False)
    return (rp - rm) / (2 * eps)


@pytest.fixture(scope="module")
def w1():
    return synth.make_config(1)


@pytest.fixture(scope="module")
def w2small():
    return synth.make_config(2, L=60, n_plane=300, n_edge=120)


def test_visual_td_jacobian_fd(oracle, w1):
    r, J = oracle.eval_factors(w1, abi.FACTOR_VISUAL)
    F = len(w1.vis_i)
    J = J.reshape(F, 46)
    r = r.reshape(F, 2)
    # columns: pose_i 2x7 | pose_j 2x7 | ex 2x7 | lam 2 | td 2
    for k in range(w1.K):
        for c in range(6):
            fd = fd_column(oracle, w1, abi.FACTOR_VISUAL, "pose", k, c).reshape(F, 2)
            ana = np.zeros((F, 2))
            mi, mj = w1.vis_i == k, w1.vis_j == k
            ana[mi] += J[mi, 0:14].reshape(-1, 2, 7)[:, :, c]
            ana[mj] += J[mj, 14:28].reshape(-1, 2, 7)[:, :, c]
            assert np.allclose(fd, ana, rtol=2e-6, atol=2e-5), (k, c, np.abs(fd - ana).max())
    for c in range(6):
        fd = fd_column(oracle, w1, abi.FACTOR_VISUAL, "ex", 0, c).reshape(F, 2)
        assert np.allclose(fd, J[:, 28:42].reshape(F, 2, 7)[:, :, c], rtol=2e-6, atol=2e-5)
    fd = fd_column(oracle, w1, abi.FACTOR_VISUAL, "td", 0, 0, eps=1e-7).reshape(F, 2)
    assert np.allclose(fd, J[:, 44:46], rtol=1e-5, atol=1e-3)
    for l in [0, 3, 77, 199]:
        fd = fd_column(oracle, w1, abi.FACTOR_VISUAL, "lam", l, 0, eps=1e-7).reshape(F, 2)
        m = w1.vis_l == l
        assert np.allclose(fd[m], J[m, 42:44], rtol=1e-5, atol=1e-3)
        assert np.all(fd[~m] == 0)
    assert np.all(J[:, 6] == 0) and np.all(J[:, 13] == 0)  # 7th column of pose blocks is zero


def test_visual_residual_numpy(oracle, w1):
    """Independent numpy re-statement of projection_td_factor.cpp:50-70 with rotation matrices."""
    r, _ = oracle.eval_factors(w1, abi.FACTOR_VISUAL, jac=False)
    r = r.reshape(-1, 2)
    ric, tic = synth.quat_to_R(w1.ex_pose[3:]), w1.ex_pose[:3]
    for f in range(0, len(w1.vis_i), 7):
        c = w1.vis_const[f]
        i, j, l = w1.vis_i[f], w1.vis_j[f], w1.vis_l[f]
        Ri, Rj = synth.quat_to_R(w1.pose[i, 3:]), synth.quat_to_R(w1.pose[j, 3:])
        td = w1.td[0]
        pi = c[0:3] - (td - c[10]) * np.array([c[6], c[7], 0])
        pj = c[3:6] - (td - c[11]) * np.array([c[8], c[9], 0])
        Xc = pi / w1.inv_depth[l]
        Xw = Ri @ (ric @ Xc + tic) + w1.pose[i, :3]
        Xcj = ric.T @ (Rj.T @ (Xw - w1.pose[j, :3]) - tic)
        ref = 230.0 * (Xcj[:2] / Xcj[2] - pj[:2])
        assert np.allclose(r[f], ref, rtol=1e-10, atol=1e-10)


def test_visual_no_td_variant(oracle, w1):
    """A7 ProjectionFactor == A6 with the td terms removed (projection_factor.cpp:21-121)."""
    w = perturbed(w1, "td", 0, 0, 0.0)
    w.use_td = 0
    r0, J0 = oracle.eval_factors(w, abi.FACTOR_VISUAL)
    w.td[0] = 0.0; w.use_td = 1
    w.vis_const = w.vis_const.copy(); w.vis_const[:, 10:12] = 0.0
    r1, J1 = oracle.eval_factors(w, abi.FACTOR_VISUAL)
    assert np.allclose(r0, r1, rtol=0, atol=1e-12)
    J0, J1 = J0.reshape(-1, 46), J1.reshape(-1, 46)
    assert np.allclose(J0[:, :44], J1[:, :44], atol=1e-12)
    assert np.all(J0[:, 44:] == 0)


def test_imu_jacobian_fd(oracle, w1):
    r, J = oracle.eval_factors(w1, abi.FACTOR_IMU)
    F = len(w1.imu_i)
    r = r.reshape(F, 15); J = J.reshape(F, 480)
    Ji, Jsi, Jj, Jsj = J[:, :105].reshape(F, 15, 7), J[:, 105:240].reshape(F, 15, 9), J[:, 240:345].reshape(F, 15, 7), J[:, 345:].reshape(F, 15, 9)
    scale = np.abs(J).max()
    for k in range(w1.K):
        for c in range(6):
            fd = fd_column(oracle, w1, abi.FACTOR_IMU, "pose", k, c).reshape(F, 15)
            ana = np.zeros((F, 15))
            ana[w1.imu_i == k] += Ji[w1.imu_i == k, :, c]
            ana[w1.imu_j == k] += Jj[w1.imu_j == k, :, c]
            assert np.abs(fd - ana).max() < 2e-6 * scale, (k, c, np.abs(fd - ana).max(), scale)
        for c in range(9):
            fd = fd_column(oracle, w1, abi.FACTOR_IMU, "sb", k, c).reshape(F, 15)
            ana = np.zeros((F, 15))
            ana[w1.imu_i == k] += Jsi[w1.imu_i == k, :, c]
            ana[w1.imu_j == k] += Jsj[w1.imu_j == k, :, c]
            # d r_theta / d bg uses delta_q instead of the corrected delta_q (imu_factor.h:127): first-order equal
            assert np.abs(fd - ana).max() < 2e-4 * scale, (k, c, np.abs(fd - ana).max(), scale)
    assert np.all(Ji[:, :, 6] == 0) and np.all(Jj[:, :, 6] == 0)


def test_imu_sqrt_info_identity(oracle, w1):
    """sqrt_info^T sqrt_info == covariance^-1 (imu_factor.h:64) and upper-triangular."""
    import ctypes as C
    cov = w1.imu_const[0, 62:].reshape(15, 15).copy()
    U = np.zeros((15, 15))
    f = oracle.lib.orc_imu_sqrt_info
    f.restype = C.c_int
    assert f(cov.ctypes.data_as(C.POINTER(C.c_double)), U.ctypes.data_as(C.POINTER(C.c_double))) == 0
    assert np.allclose(np.tril(U, -1), 0)
    info = np.linalg.inv(cov)
    assert np.allclose(U.T @ U, info, rtol=1e-7, atol=1e-7 * np.abs(info).max())


def test_imu_residual_small_at_truth(oracle, w1):
    """With the true states the whitened IMU residual is O(1) per component (noise-consistent)."""
    w = perturbed(w1, "td", 0, 0, 0.0)
    w.pose, w.speedbias = w1.truth["pose"].copy(), w1.truth["speedbias"].copy()
    r, _ = oracle.eval_factors(w, abi.FACTOR_IMU, jac=False)
    assert np.sqrt(np.mean(r ** 2)) < 5.0


def test_lidar_point_factors_fd_and_reference_form(oracle, w2small):
    import ctypes as C
    w = w2small
    dp = C.POINTER(C.c_double)
    for cls, nr in ((abi.FACTOR_EDGE, 3), (abi.FACTOR_PLANE, 1)):
        r, J = oracle.eval_factors(w, cls)
        F = w.nfactors(cls)
        J = J.reshape(F, nr, 7); r = r.reshape(F, nr)
        pose_of = w.edge_pose if cls == abi.FACTOR_EDGE else w.plane_pose
        for k in range(w.K):
            for c in range(6):
                fd = fd_column(oracle, w, cls, "pose", k, c).reshape(F, nr)
                ana = np.where((pose_of == k)[:, None], J[:, :, c], 0.0)
                assert np.allclose(fd, ana, rtol=1e-6, atol=1e-6), (cls, k, c, np.abs(fd - ana).max())
        assert np.all(J[:, :, 6] == 0)
        # literal lidarFactor.hpp functor on the composed LiDAR->world transform gives the same residual
        Rbl = synth.RLB.T; tbl = -synth.RLB.T @ synth.TLB
        for f in range(0, F, 11):
            k = pose_of[f]
            Rk = synth.quat_to_R(w.pose[k, 3:])
            Rwl = Rk @ Rbl; twl = Rk @ tbl + w.pose[k, :3]
            q = synth.R_to_quat(Rwl); out = np.zeros(3)
            if cls == abi.FACTOR_EDGE:
                c9 = w.edge_const[f]
                oracle.lib.orc_edge_residual_ref(abi.f64(c9[0:3]).ctypes.data_as(dp), abi.f64(c9[3:6]).ctypes.data_as(dp), abi.f64(c9[6:9]).ctypes.data_as(dp),
                                                 q.ctypes.data_as(dp), twl.ctypes.data_as(dp), out.ctypes.data_as(dp))
                assert np.allclose(out, r[f], atol=1e-9)
            else:
                c7 = w.plane_const[f]
                oracle.lib.orc_plane_residual_ref(abi.f64(c7[0:3]).ctypes.data_as(dp), abi.f64(c7[3:6]).ctypes.data_as(dp), C.c_double(c7[6]),
                                                  q.ctypes.data_as(dp), twl.ctypes.data_as(dp), out.ctypes.data_as(dp))
                assert np.allclose(out[0], r[f, 0], atol=1e-9)


def _fd_global(oracle, w, cls, k, col, eps=1e-6):
    """Raw d/d(global coordinate): AutoDiff factors differentiate w.r.t. [p, qx, qy, qz, qw] directly."""
    wp, wm = perturbed(w, "td", 0, 0, 0.0), perturbed(w, "td", 0, 0, 0.0)
    wp.pose[k, col] += eps; wm.pose[k, col] -= eps
    rp, _ = oracle.eval_factors(wp, cls, jac=False)
    rm, _ = oracle.eval_factors(wm, cls, jac=False)
    return (rp - rm) / (2 * eps)


def test_icp_lps_autodiff_matches_raw_fd(oracle, w2small):
    w = w2small
    for cls, nb, ids in ((abi.FACTOR_ICP, 4, w.icp_ids), (abi.FACTOR_LPS, 2, w.lps_ids)):
        r, J = oracle.eval_factors(w, cls)
        F = len(ids)
        J = J.reshape(F, nb, 3, 7)
        for k in range(w.K):
            for col in range(7):
                fd = _fd_global(oracle, w, cls, k, col).reshape(F, 3)
                ana = np.zeros((F, 3))
                for f in range(F):
                    for b in range(nb):
                        if ids[f, b] == k:
                            ana[f] += J[f, b, :, col]
                assert np.allclose(fd, ana, rtol=1e-6, atol=1e-5), (cls, k, col, np.abs(fd - ana).max())
    r, _ = oracle.eval_factors(w, abi.FACTOR_ICP, jac=False)
    assert np.all(r.reshape(-1, 3)[:, 1] == 0)  # y component forced to zero (lidar_backend.h:157)


def test_lps_residual_numpy(oracle, w2small):
    w = w2small
    r, _ = oracle.eval_factors(w, abi.FACTOR_LPS, jac=False)
    r = r.reshape(-1, 3)
    for f in range(len(w.lps_ids)):
        tl, tr, tk = w.lps_const[f, :3]
        qi = synth.slerp(w.pose[w.lps_ids[f, 0], 3:], w.pose[w.lps_ids[f, 1], 3:], (tk - tl) / (tr - tl))
        qinv = np.array([-qi[0], -qi[1], -qi[2], qi[3]]) / np.dot(qi, qi)
        q12 = synth.qmul(qinv, w.lps_const[f, 3:7])
        assert np.allclose(r[f], 2 * q12[:3] / 0.01, atol=1e-9)


def test_prior_factor(oracle, w2small):
    w = w2small
    pr = w.prior
    assert pr.n == 6 * w.K + 10
    r, J = oracle.eval_factors(w, abi.FACTOR_PRIOR)
    Jm = pr.J_matrix()
    # dx by hand (marginalization_factor.cpp:362-382)
    dx = np.zeros(pr.n); xo = 0
    for b in range(len(pr.blk_kind)):
        kind, idx, col = int(pr.blk_kind[b]), int(pr.blk_index[b]), int(pr.blk_col[b])
        gs = {0: 7, 1: 9, 2: 7, 3: 1}[kind]
        x = {0: lambda: w.pose[idx], 1: lambda: w.speedbias[idx], 2: lambda: w.ex_pose, 3: lambda: w.td}[kind]()
        x0 = pr.x0[xo:xo + gs]; xo += gs
        if gs != 7:
            dx[col:col + gs] = x - x0
        else:
            dx[col:col + 3] = x[:3] - x0[:3]
            q0inv = np.array([-x0[3], -x0[4], -x0[5], x0[6]]) / np.dot(x0[3:], x0[3:])
            dq = synth.qmul(q0inv, x[3:])
            dx[col + 3:col + 6] = (2.0 if dq[3] >= 0 else -2.0) * dq[:3]
    assert np.allclose(r, pr.r0 + Jm @ dx, rtol=1e-12, atol=1e-12)
    off = 0
    for b in range(len(pr.blk_kind)):
        kind, col = int(pr.blk_kind[b]), int(pr.blk_col[b])
        gs, ls = {0: (7, 6), 1: (9, 9), 2: (7, 6), 3: (1, 1)}[kind]
        blk = J[off:off + pr.n * gs].reshape(pr.n, gs); off += pr.n * gs
        assert np.array_equal(blk[:, :ls], Jm[:, col:col + ls])
        assert np.all(blk[:, ls:] == 0)


def test_loss_functions(oracle):
    import ctypes as C
    rho = np.zeros(3)
    for s in (0.0, 0.3, 1.0, 7.5, 400.0):
        oracle.lib.orc_loss(C.c_int(abi.LOSS_CAUCHY), C.c_double(1.0), C.c_double(s), rho.ctypes.data_as(C.POINTER(C.c_double)))
        assert np.allclose(rho, [np.log1p(s), 1 / (1 + s), -1 / (1 + s) ** 2])
        oracle.lib.orc_loss(C.c_int(abi.LOSS_HUBER), C.c_double(0.1), C.c_double(s), rho.ctypes.data_as(C.POINTER(C.c_double)))
        if s <= 0.01:
            assert np.allclose(rho, [s, 1, 0])
        else:
            assert np.allclose(rho, [2 * 0.1 * np.sqrt(s) - 0.01, 0.1 / np.sqrt(s), -0.1 / np.sqrt(s) / (2 * s)])


def test_preintegration_matches_numpy(oracle):
    """oracle's C++ restatement of integration_base.h:54-158 vs the generator's numpy restatement."""
    import ctypes as C
    rng = np.random.default_rng(5)
    n = 20
    dts = np.full(n, 0.005)
    acc = rng.normal(0, 1, (n + 1, 3)) + [0, 0, 9.8]
    gyr = rng.normal(0, 0.3, (n + 1, 3))
    ba, bg = rng.normal(0, 0.02, 3), rng.normal(0, 0.002, 3)
    ref = synth.preintegrate(dts, acc[1:], gyr[1:], acc[0], gyr[0], ba, bg)
    out = np.zeros(287)
    noise = np.array([synth.ACC_N, synth.GYR_N, synth.ACC_W, synth.GYR_W])
    dp = C.POINTER(C.c_double)
    a1, g1 = np.ascontiguousarray(acc[1:]), np.ascontiguousarray(gyr[1:])
    oracle.lib.orc_preintegrate(C.c_int(n), dts.ctypes.data_as(dp), a1.ctypes.data_as(dp), g1.ctypes.data_as(dp),
                                np.ascontiguousarray(acc[0]).ctypes.data_as(dp), np.ascontiguousarray(gyr[0]).ctypes.data_as(dp),
                                ba.ctypes.data_as(dp), bg.ctypes.data_as(dp), noise.ctypes.data_as(dp), out.ctypes.data_as(dp))
    assert np.allclose(out, ref, rtol=1e-10, atol=1e-14)


# ---- the four functors of lidar_mapping/src/lidarFactor.hpp at one pose (vil_eval_lidar_functors / orc_eval_lidar_functors) --------
def _functor_case(seed=5, n=40):
    rng = np.random.default_rng(seed)
    q_lb = rng.normal(size=4); q_lb /= np.linalg.norm(q_lb)
    t_lb = rng.normal(size=3) * 0.1
    pose = np.concatenate([rng.normal(size=3), (lambda q: q / np.linalg.norm(q))(rng.normal(size=4))])
    cp = rng.normal(size=(n, 3)) * 5
    return rng, q_lb, t_lb, pose, cp


def _lidar_to_world(q_lb, t_lb, pose, cp):
    Rlb = synth.quat_to_R(q_lb); R = synth.quat_to_R(pose[3:])
    pb = (cp - t_lb) @ Rlb                      # R_lb^T (p_l - t_lb)
    return pb @ R.T + pose[:3], pb, R


def eval_functors(be, kind, consts, q_lb, t_lb, pose, nr):
    import ctypes as C
    n = len(consts)
    c = np.ascontiguousarray(consts, dtype=np.float64)
    r = np.zeros(n * nr); J = np.zeros(n * nr * 7)
    dp = C.POINTER(C.c_double)
    f = getattr(be.lib, be.prefix + "eval_lidar_functors"); f.restype = C.c_int
    args = (C.c_int32(kind), C.c_int32(n), c.ctypes.data_as(dp), abi.f64(q_lb).ctypes.data_as(dp), abi.f64(t_lb).ctypes.data_as(dp), abi.f64(pose).ctypes.data_as(dp), r.ctypes.data_as(dp), J.ctypes.data_as(dp))
    st = f(be.ctx, *args) if be.has_ctx else f(*args)
    assert st == 0
    return r.reshape(n, nr), J.reshape(n, nr, 7)


def functor_tables():
    rng, q_lb, t_lb, pose, cp = _functor_case()
    n = len(cp)
    j, l, m = rng.normal(size=(n, 3)) * 4, rng.normal(size=(n, 3)) * 4, rng.normal(size=(n, 3)) * 4
    closed = rng.normal(size=(n, 3)) * 5
    return q_lb, t_lb, pose, cp, np.hstack([cp, j, l, m]), np.hstack([cp, closed])


def test_plane3_and_distance_functors_match_the_literal_functors(oracle):
    """LidarPlaneFactor (lidarFactor.hpp:57-104, s = 1: slerp(1) is the quaternion itself) and LidarDistanceFactor (:141-172)
    re-derived in numpy on the LiDAR->world transform, and their Jacobians by the reference's finite-difference recipe."""
    q_lb, t_lb, pose, cp, c12, c6 = functor_tables()
    pw, pb, R = _lidar_to_world(q_lb, t_lb, pose, cp)
    r3, J3 = eval_functors(oracle, 1, c12, q_lb, t_lb, pose, 1)
    nrm = np.cross(c12[:, 3:6] - c12[:, 6:9], c12[:, 3:6] - c12[:, 9:12]); nrm /= np.linalg.norm(nrm, axis=1)[:, None]
    assert np.abs(r3[:, 0] - np.einsum("ij,ij->i", pw - c12[:, 3:6], nrm)).max() < 1e-12
    rd, Jd = eval_functors(oracle, 3, c6, q_lb, t_lb, pose, 3)
    assert np.abs(rd - (pw - c6[:, 3:6])).max() < 1e-12
    for kind, c, nr, ana in ((1, c12, 1, J3), (3, c6, 3, Jd)):
        for col in range(6):
            d = np.zeros(6); d[col] = 1e-6
            rp, _ = eval_functors(oracle, kind, c, q_lb, t_lb, plus_pose(pose, d), nr)
            rm, _ = eval_functors(oracle, kind, c, q_lb, t_lb, plus_pose(pose, -d), nr)
            assert np.allclose((rp - rm) / 2e-6, ana[:, :, col], rtol=1e-6, atol=1e-6)
        assert np.all(ana[:, :, 6] == 0.0)


def test_edge_and_plane_norm_functor_batches_equal_the_window_classes(oracle):
    """kinds 0 / 2 are the window's EDGE / PLANE classes evaluated at one pose."""
    w = synth.make_config(2, L=20, n_plane=60, n_edge=40)
    k = 3
    pm, em = w.plane_pose == k, w.edge_pose == k
    rP, JP = oracle.eval_factors(w, abi.FACTOR_PLANE); rE, JE = oracle.eval_factors(w, abi.FACTOR_EDGE)
    r2, J2 = eval_functors(oracle, 2, w.plane_const[pm], w.q_lb, w.t_lb, w.pose[k], 1)
    r0, J0 = eval_functors(oracle, 0, w.edge_const[em], w.q_lb, w.t_lb, w.pose[k], 3)
    assert np.array_equal(r2[:, 0], rP[pm]) and np.array_equal(J2.reshape(-1, 7), JP.reshape(-1, 7)[pm])
    assert np.array_equal(r0, rE.reshape(-1, 3)[em]) and np.array_equal(J0.reshape(-1, 21), JE.reshape(-1, 21)[em])

"""Plain-numpy restatement of the normal equations / marginalisation used as an INDEPENDENT pin (test infrastructure).

Built only on per-factor residuals and Evaluate()-layout Jacobians handed in by a backend's eval_factors (the CPU oracle
in the -m "not gpu" suite, the HIP library in the -m gpu suite): the loss corrector of ResidualBlockInfo::Evaluate
(marginalization_factor.cpp:37-67), the "first local_size columns" rule (:141-174), dense normal equations over ALL
parameters, dense Schur complements (numpy.linalg eigen pseudo-inverse = the reference's rule, :273-290; mpmath at 60
digits = the exact value the rule approximates).  Shares no code with oracle_solver.cpp / oracle_marg.cpp / the kernels.
"""
import numpy as np

from mvil_fusion_amd import abi, synth

GS = {abi.BLK_POSE: 7, abi.BLK_SPEEDBIAS: 9, abi.BLK_EX: 7, abi.BLK_TD: 1}


def loss_rho(kind, a, s):
    """ceres::CauchyLoss / HuberLoss / none: rho, rho', rho'' at s."""
    if kind == abi.LOSS_CAUCHY:
        b = a * a
        c = 1.0 / b
        sm = 1.0 + s * c
        inv = 1.0 / sm
        return b * np.log(sm), max(np.finfo(float).tiny, inv), -c * inv * inv
    if kind == abi.LOSS_HUBER:
        b = a * a
        if s > b:
            r = np.sqrt(s)
            rho1 = max(np.finfo(float).tiny, a / r)
            return 2.0 * a * r - b, rho1, -rho1 / (2.0 * s)
        return s, 1.0, 0.0
    return s, 1.0, 0.0


def corrected(kind, a, r, Js):
    """marginalization_factor.cpp:37-67: returns (0.5 rho, corrected r, corrected Jacobian blocks)."""
    if kind == abi.LOSS_NONE:
        return 0.5 * float(r @ r), r, Js
    sq = float(r @ r)
    rho0, rho1, rho2 = loss_rho(kind, a, sq)
    s1 = np.sqrt(rho1)
    if sq == 0.0 or rho2 <= 0.0:
        scal, asq = s1, 0.0
    else:
        Dd = 1.0 + 2.0 * sq * rho2 / rho1
        alpha = 1.0 - np.sqrt(Dd)
        scal, asq = s1 / (1.0 - alpha), alpha / sq
    Jc = [s1 * (J - asq * np.outer(r, r @ J)) for J in Js]
    return 0.5 * rho0, r * scal, Jc


def factor_list(oracle, w, opts):
    """Every residual block of the window as (cost, r, [(key, J_local)]) with key = ('pose', k) / ('sb', k) / ('ex',) / ('td',) / ('lam', l);
    J_local = the first local_size columns of the Evaluate() block (pose 7 -> 6)."""
    out = []
    K = w.K

    def loc(J, gs):
        return J[:, :6] if gs == 7 else J

    r, J = oracle.eval_factors(w, abi.FACTOR_IMU)
    for f in range(len(w.imu_i)):
        if w.imu_const[f, 16] > 10.0:        # estimator.cpp:1182
            continue
        i, j = int(w.imu_i[f]), int(w.imu_j[f])
        Jf = J[480 * f: 480 * (f + 1)]
        blocks = [(("pose", i), loc(Jf[0:105].reshape(15, 7), 7)), (("sb", i), Jf[105:240].reshape(15, 9)),
                  (("pose", j), loc(Jf[240:345].reshape(15, 7), 7)), (("sb", j), Jf[345:480].reshape(15, 9))]
        out.append((abi.LOSS_NONE, 0.0, r[15 * f: 15 * f + 15], blocks))
    r, J = oracle.eval_factors(w, abi.FACTOR_VISUAL)
    for f in range(len(w.vis_i)):
        i, j, l = int(w.vis_i[f]), int(w.vis_j[f]), int(w.vis_l[f])
        Jf = J[46 * f: 46 * (f + 1)]
        blocks = [(("pose", i), loc(Jf[0:14].reshape(2, 7), 7)), (("pose", j), loc(Jf[14:28].reshape(2, 7), 7)),
                  (("ex",), loc(Jf[28:42].reshape(2, 7), 7)), (("lam", l), Jf[42:44].reshape(2, 1))]
        if w.use_td:
            blocks.append((("td",), Jf[44:46].reshape(2, 1)))
        out.append((opts.visual_loss, opts.visual_loss_scale, r[2 * f: 2 * f + 2], blocks))
    r, J = oracle.eval_factors(w, abi.FACTOR_ICP)
    for f in range(len(w.icp_ids)):
        Jf = J[84 * f: 84 * (f + 1)]
        blocks = [(("pose", int(w.icp_ids[f, b])), loc(Jf[21 * b: 21 * b + 21].reshape(3, 7), 7)) for b in range(4)]
        out.append((opts.rel_loss, opts.rel_loss_scale, r[3 * f: 3 * f + 3], blocks))
    r, J = oracle.eval_factors(w, abi.FACTOR_LPS)
    for f in range(len(w.lps_ids)):
        Jf = J[42 * f: 42 * (f + 1)]
        blocks = [(("pose", int(w.lps_ids[f, b])), loc(Jf[21 * b: 21 * b + 21].reshape(3, 7), 7)) for b in range(2)]
        out.append((opts.rel_loss, opts.rel_loss_scale, r[3 * f: 3 * f + 3], blocks))
    r, J = oracle.eval_factors(w, abi.FACTOR_EDGE)
    for f in range(len(w.edge_pose)):
        out.append((opts.lidar_loss, opts.lidar_loss_scale, r[3 * f: 3 * f + 3], [(("pose", int(w.edge_pose[f])), loc(J[21 * f: 21 * f + 21].reshape(3, 7), 7))]))
    r, J = oracle.eval_factors(w, abi.FACTOR_PLANE)
    for f in range(len(w.plane_pose)):
        out.append((opts.lidar_loss, opts.lidar_loss_scale, r[f: f + 1], [(("pose", int(w.plane_pose[f])), loc(J[7 * f: 7 * f + 7].reshape(1, 7), 7))]))
    if w.prior.n:
        r, J = oracle.eval_factors(w, abi.FACTOR_PRIOR)
        n, blocks, jo = w.prior.n, [], 0
        for b in range(len(w.prior.blk_kind)):
            kind, idx = int(w.prior.blk_kind[b]), int(w.prior.blk_index[b])
            gs = GS[kind]
            key = {abi.BLK_POSE: ("pose", idx), abi.BLK_SPEEDBIAS: ("sb", idx), abi.BLK_EX: ("ex",), abi.BLK_TD: ("td",)}[kind]
            blocks.append((key, loc(J[jo: jo + n * gs].reshape(n, gs), gs)))
            jo += n * gs
        out.append((abi.LOSS_NONE, 0.0, r, blocks))
    assert K == w.K
    return out


def normal_equations(facs, index):
    """H = sum J^T J, b = sum J^T r over the corrected blocks; index: key -> first column (absent key = constant block)."""
    N = max(v + {"pose": 6, "sb": 9, "ex": 6, "td": 1, "lam": 1}[k[0]] for k, v in index.items())
    H, g, cost = np.zeros((N, N)), np.zeros(N), 0.0
    for kind, a, r, blocks in facs:
        c, rc, Jc = corrected(kind, a, np.asarray(r, float), [B for _, B in blocks])
        cost += c
        for (ka, _), Ja in zip(blocks, Jc):
            if ka not in index:
                continue
            ia = index[ka]
            g[ia: ia + Ja.shape[1]] += Ja.T @ rc
            for (kb, _), Jb in zip(blocks, Jc):
                if kb in index:
                    ib = index[kb]
                    H[ia: ia + Ja.shape[1], ib: ib + Jb.shape[1]] += Ja.T @ Jb
    return cost, H, g


def camera_index(w, free_only=True):
    K = w.K
    idx = {}
    for k in range(K):
        if not (free_only and w.pose_const[k]):
            idx[("pose", k)] = 6 * k
        if not (free_only and w.sb_const[k]):
            idx[("sb", k)] = 6 * K + 7 + 9 * k
    if not (free_only and w.ex_const):
        idx[("ex",)] = 6 * K
    if w.use_td and not (free_only and w.td_const):
        idx[("td",)] = 6 * K + 6
    return idx


def reduced_system(backend, w, opts):
    """cost, S (D x D), g (D): numpy normal equations over every free parameter, landmarks eliminated by a dense Schur complement."""
    D, L = w.D, w.L
    index = camera_index(w)
    obs = np.bincount(w.vis_l, minlength=L) if L else np.zeros(0, int)
    for l in range(L):
        if not w.lm_const[l] and obs[l] > 0:
            index[("lam", l)] = D + l
    facs = factor_list(backend, w, opts)
    N = D + L
    cost, H, g = normal_equations(facs, index)
    H = np.pad(H, ((0, N - H.shape[0]), (0, N - H.shape[1])))
    g = np.pad(g, (0, N - g.shape[0]))
    Hcc, Hcl, hll, bc, bl = H[:D, :D], H[:D, D:], np.diag(H[D:, D:]).copy(), g[:D], g[D:]
    assert np.abs(H[D:, D:] - np.diag(hll)).max() == 0.0          # landmarks never share a factor
    live = hll != 0.0
    ip = np.where(live, 1.0 / np.where(live, hll, 1.0), 0.0)
    return cost, Hcc - (Hcl * ip) @ Hcl.T, bc - Hcl @ (ip * bl)


def rank_deficient_window(backend_for_prior, offset=2e-7):
    """configs[1]-mini window evaluated at a state where frame 1 sits (almost) on top of frame 0: every landmark anchored in
    frame 0 and seen from frame 1 only has no parallax, its h_ll = J_l^T J_l drops to ~1e-9 -- below the eps = 1e-8 of
    marginalization_factor.cpp:277 -- while the same factors still constrain the rotations strongly."""
    w = synth.make_config(2, L=120, n_plane=0, n_edge=0, prior_fn=lambda pre: backend_for_prior.marginalize(pre).to_prior())
    w.pose[1] = w.pose[0].copy()
    w.pose[1][0] += offset
    obs = np.bincount(w.vis_l, minlength=w.L)
    weak = [int(l) for l in np.unique(w.vis_l[(w.vis_i == 0) & (w.vis_j == 1)]) if obs[l] == 1]
    return w, weak


def marg_numpy(oracle, w, opts, flag, icp_marg=-1, lps_marg=-1, eps=1e-8, lidar=False):
    """estimator.cpp:1486-1616 / 1624-1681 + marginalization_factor.cpp:176-290 in numpy (fp64, eigh): dict with the kept keys, m,
    A, b, the eigenvalues of A_mm and the full (dropped | kept) system.  lidar=True also collects the LiDAR point factors of the
    dropped pose (the extended mode of this build)."""
    K = w.K
    wf = synth.make_config.__globals__["Window"](w.K, w.L)
    wf.__dict__.update({k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in w.__dict__.items()})
    wf.pose_const[:] = 0; wf.sb_const[:] = 0; wf.lm_const[:] = 0; wf.ex_const = 0; wf.td_const = 0
    allf = factor_list(oracle, wf, opts)
    old = flag == abi.MARGIN_OLD
    drop_pose = 0 if old else K - 2
    sel = []
    # which residual blocks MarginalizationInfo collects
    n_imu = sum(1 for f in range(len(w.imu_i)) if w.imu_const[f, 16] <= 10.0)
    imu_rows = [f for f in range(len(w.imu_i)) if w.imu_const[f, 16] <= 10.0]
    nvis, nicp, nlps = len(w.vis_i), len(w.icp_ids), len(w.lps_ids)
    o_imu, o_vis, o_icp, o_lps = 0, n_imu, n_imu + nvis, n_imu + nvis + nicp
    o_edge = o_lps + nlps
    o_prior = o_edge + len(w.edge_pose) + len(w.plane_pose)
    if w.prior.n:
        keys = [k for k, _ in allf[o_prior][3]]
        if old or ("pose", drop_pose) in keys:
            sel.append(allf[o_prior])
        elif not old:
            return None
    elif not old:
        return None
    drop = []
    if old:
        drop += [("pose", 0), ("sb", 0)]
        for q, f in enumerate(imu_rows):
            if w.imu_i[f] == 0 and w.imu_j[f] == 1 and w.imu_const[f, 16] < 10.0:
                sel.append(allf[o_imu + q])
        for f in range(nvis):
            if w.vis_i[f] == 0:
                sel.append(allf[o_vis + f])
                if ("lam", int(w.vis_l[f])) not in drop:
                    drop.append(("lam", int(w.vis_l[f])))
        if icp_marg >= 0:
            sel.append(allf[o_icp + icp_marg])
        if lps_marg >= 0:
            sel.append(allf[o_lps + lps_marg])
        if lidar:       # LiDAR point factors of the dropped pose (they touch nothing else)
            for f in range(len(w.edge_pose)):
                if w.edge_pose[f] == 0:
                    sel.append(allf[o_edge + f])
            for f in range(len(w.plane_pose)):
                if w.plane_pose[f] == 0:
                    sel.append(allf[o_edge + len(w.edge_pose) + f])
    else:
        drop.append(("pose", drop_pose))
    used = []
    for _, _, _, blocks in sel:
        for k, _ in blocks:
            if k not in used:
                used.append(k)
    drop = [k for k in drop if k in used]
    rank = {"pose": 0, "sb": 1, "ex": 2, "td": 3}
    kept = sorted([k for k in used if k not in drop], key=lambda k: (rank[k[0]], k[1] if len(k) > 1 else 0))
    size = {"pose": 6, "sb": 9, "ex": 6, "td": 1, "lam": 1}
    index, p = {}, 0
    for k in drop + kept:
        index[k] = p
        p += size[k[0]]
    m = sum(size[k[0]] for k in drop)
    _, A, b = normal_equations(sel, index)
    Amm = 0.5 * (A[:m, :m] + A[:m, :m].T)
    wv, V = np.linalg.eigh(Amm)
    inv = np.where(wv > eps, 1.0 / np.where(wv > eps, wv, 1.0), 0.0)
    Ainv = (V * inv) @ V.T
    Ar = A[m:, m:] - A[m:, :m] @ Ainv @ A[:m, m:]
    br = b[m:] - A[m:, :m] @ Ainv @ b[:m]
    return dict(kept=kept, drop=drop, m=m, A=Ar, b=br, eig_mm=wv, A_full=A, b_full=b)


def exact_schur(A_full, b_full, m, eps=1e-8, dps=60):
    """The Schur complement A_rr - A_rm pinv_eps(A_mm) A_mr (and b) of marginalization_factor.cpp:273-290 evaluated with
    60-digit arithmetic (mpmath): what the reference's fp64 eigen-decomposition approximates.  With cond(A_mm) ~ 1e7 and the
    cancellation against the 1e10-1e11 bias information, the fp64 eigen route is itself only good to ~1e-6 (diagonally scaled)."""
    import mpmath as mp
    with mp.workdps(dps):
        Amm = mp.matrix((0.5 * (A_full[:m, :m] + A_full[:m, :m].T)).tolist())
        Amr, Arm, Arr = mp.matrix(A_full[:m, m:].tolist()), mp.matrix(A_full[m:, :m].tolist()), mp.matrix(A_full[m:, m:].tolist())
        bm, br = mp.matrix(b_full[:m].tolist()), mp.matrix(b_full[m:].tolist())
        E, Q = mp.eigsy(Amm)
        inv = mp.matrix(m, m)
        for k in range(m):
            if E[k] > eps:
                inv[k, k] = 1 / E[k]
        Ainv = Q * inv * Q.T
        Ax = Arr - Arm * Ainv * Amr
        bx = br - Arm * Ainv * bm
        return np.array(Ax.tolist(), dtype=float), np.array(bx.tolist(), dtype=float).ravel(), np.array([float(e) for e in E])


def scaled_err(A, B):
    sc = np.sqrt(np.maximum(np.abs(np.diag(B)), 1e-300))
    return np.abs((A - B) / np.outer(sc, sc)).max()



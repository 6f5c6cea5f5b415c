"""Deterministic synthetic sliding windows for the BASELINE.json configs.

The reference's data (3indoor.bag / outdoor.bag, README.md:21-28) is not available offline, so every
window is synthesised: a figure-eight trajectory sampled at 10 Hz keyframes, a 200 Hz IMU stream
pre-integrated with the mid-point scheme of integration_base.h:54-158 (restated here in numpy --
this module is product-side input generation and must not touch oracle/), pinhole observations of
landmarks with the track-length rule of SURVEY 8d, LiDAR plane/edge points of a 20x20x5 m room built
with the correspondence recipe of lidar_mapping/src/localMapping.cpp:651-662,724-726, a few
scan-to-scan ICP constraints and LPS rotation priors.

RNG: numpy Generator(PCG64(seed)), seed = 20240601 + config_id.
"""
import numpy as np

from .abi import Window, Prior, BLK_POSE, BLK_SPEEDBIAS, BLK_EX, BLK_TD, VIL_IMU_CONST

ACC_N, GYR_N, ACC_W, GYR_W = 0.02065, 0.00519, 0.00667, 0.00088056  # yaml:81-86
G_NORM = 9.795                                                       # yaml:102
FOCAL_LENGTH = 460.0                                                 # parameters.h:11
RIC = np.array([[0.99999072, -0.00209387, -0.00376471], [-0.00208308, -0.99999371, 0.0028693], [-0.0037707, -0.00286143, -0.9999888]])  # yaml:31-36
TIC = np.array([-0.04571386, 0.01268073, -0.01535602])               # yaml:38-42
RLB_RAW = np.array([[-0.0320631, 0.000946093, -0.999485], [-0.999482, -0.00274554, 0.0320604], [-0.0027138, 0.999996, 0.00103363]])  # yaml gt_rli
TLB = np.array([0.2, -0.005, -0.1])                                  # yaml gt_tli
KF_DT, IMU_DT = 0.1, 0.005

CONFIGS = {
    # id: K, L, n_plane, n_edge, n_icp, n_lps   (BASELINE.json configs[0..3]; C5 = replay, see replay.py)
    1: dict(K=5, L=200, n_plane=0, n_edge=0, n_icp=0, n_lps=0, prior=False),
    2: dict(K=10, L=1000, n_plane=24000, n_edge=6000, n_icp=3, n_lps=4, prior=True),
    3: dict(K=10, L=4000, n_plane=96000, n_edge=24000, n_icp=3, n_lps=4, prior=True),
    4: dict(K=20, L=2000, n_plane=0, n_edge=0, n_icp=3, n_lps=4, prior=True),
}


# ---- small SO(3) helpers ---------------------------------------------------------------------------
def quat_to_R(q):  # q = [x y z w]
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def R_to_quat(R):
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    else:
        i = int(np.argmax(np.diag(R))); j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
        q = np.zeros(4)
        q[i] = 0.25 * s; q[3] = (R[k, j] - R[j, k]) / s; q[j] = (R[j, i] + R[i, j]) / s; q[k] = (R[k, i] + R[i, k]) / s
    return q / np.linalg.norm(q)


def qmul(a, b):  # [x y z w]
    ax, ay, az, aw = a; bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx, aw * bw - ax * bx - ay * by - az * bz])


def _orthonormalise(R):
    return quat_to_R(R_to_quat(R))


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])


def expm_so3(w):
    th = np.linalg.norm(w)
    if th < 1e-12:
        return np.eye(3) + skew(w)
    K = skew(w / th)
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def slerp(qa, qb, t):
    d = float(np.dot(qa, qb))
    if abs(d) >= 1 - 2.220446049250313e-16:
        s0, s1 = 1 - t, t
    else:
        th = np.arccos(abs(d)); s0, s1 = np.sin((1 - t) * th) / np.sin(th), np.sin(t * th) / np.sin(th)
    if d < 0:
        s1 = -s1
    return s0 * qa + s1 * qb


RLB = _orthonormalise(RLB_RAW)   # the yaml matrices are rounded to ~6 digits
RIC = _orthonormalise(RIC)


# ---- trajectory -------------------------------------------------------------------------------------
class Trajectory:
    """Figure-eight, ~1 m/s, yaw rate <= 0.25 rad/s, gentle roll/pitch."""

    def __init__(self, phase=0.0):
        self.ph = phase

    def p(self, t):
        t = t + self.ph
        return np.array([4.0 * np.sin(0.25 * t), 2.0 * np.sin(0.5 * t), 0.3 * np.sin(0.4 * t)])

    def v(self, t):
        t = t + self.ph
        return np.array([1.0 * np.cos(0.25 * t), 1.0 * np.cos(0.5 * t), 0.12 * np.cos(0.4 * t)])

    def a(self, t):
        t = t + self.ph
        return np.array([-0.25 * np.sin(0.25 * t), -0.5 * np.sin(0.5 * t), -0.048 * np.sin(0.4 * t)])

    def R(self, t):
        t = t + self.ph
        yaw, pitch, roll = 0.5 * np.sin(0.5 * t) + 0.3, 0.08 * np.sin(0.7 * t), 0.06 * np.cos(0.9 * t)
        cy, sy, cp, sp, cr, sr = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
        Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1.0]]); Ry = np.array([[cp, 0, sp], [0, 1.0, 0], [-sp, 0, cp]]); Rx = np.array([[1.0, 0, 0], [0, cr, -sr], [0, sr, cr]])
        return Rz @ Ry @ Rx

    def w_body(self, t, h=1e-5):
        dR = (self.R(t + h) - self.R(t - h)) / (2 * h)
        W = self.R(t).T @ dR
        return np.array([W[2, 1] - W[1, 2], W[0, 2] - W[2, 0], W[1, 0] - W[0, 1]]) * 0.5


# ---- mid-point pre-integration (integration_base.h:54-158) in numpy ------------------------------------
def preintegrate(dts, accs, gyrs, acc0, gyr0, ba, bg):
    dp, dv, dq = np.zeros(3), np.zeros(3), np.array([0, 0, 0, 1.0])
    J, P = np.eye(15), np.zeros((15, 15))
    N = np.diag(np.repeat([ACC_N ** 2, GYR_N ** 2, ACC_N ** 2, GYR_N ** 2, ACC_W ** 2, GYR_W ** 2], 3))
    a0, g0, sum_dt = np.array(acc0, float), np.array(gyr0, float), 0.0
    for dt, a1, g1 in zip(dts, accs, gyrs):
        Rd = quat_to_R(dq)
        un_acc_0 = Rd @ (a0 - ba)
        un_gyr = 0.5 * (g0 + g1) - bg
        rq = qmul(dq, np.array([un_gyr[0] * dt / 2, un_gyr[1] * dt / 2, un_gyr[2] * dt / 2, 1.0]))
        Rr = quat_to_R(rq)  # Eigen toRotationMatrix() of the un-normalised product
        un_acc_1 = _qrot(rq, a1 - ba)
        un_acc = 0.5 * (un_acc_0 + un_acc_1)
        rp = dp + dv * dt + 0.5 * un_acc * dt * dt
        rv = dv + un_acc * dt
        Rw, Ra0, Ra1 = skew(un_gyr), skew(a0 - ba), skew(a1 - ba)
        I3 = np.eye(3)
        F = np.zeros((15, 15)); V = np.zeros((15, 18))
        F[0:3, 0:3] = I3
        F[0:3, 3:6] = -0.25 * Rd @ Ra0 * dt * dt + -0.25 * Rr @ Ra1 @ (I3 - Rw * dt) * dt * dt
        F[0:3, 6:9] = I3 * dt
        F[0:3, 9:12] = -0.25 * (Rd + Rr) * dt * dt
        F[0:3, 12:15] = -0.25 * Rr @ Ra1 * dt * dt * -dt
        F[3:6, 3:6] = I3 - Rw * dt
        F[3:6, 12:15] = -I3 * dt
        F[6:9, 3:6] = -0.5 * Rd @ Ra0 * dt + -0.5 * Rr @ Ra1 @ (I3 - Rw * dt) * dt
        F[6:9, 6:9] = I3
        F[6:9, 9:12] = -0.5 * (Rd + Rr) * dt
        F[6:9, 12:15] = -0.5 * Rr @ Ra1 * dt * -dt
        F[9:12, 9:12] = I3; F[12:15, 12:15] = I3
        V[0:3, 0:3] = 0.25 * Rd * dt * dt
        V[0:3, 3:6] = 0.25 * -Rr @ Ra1 * dt * dt * 0.5 * dt
        V[0:3, 6:9] = 0.25 * Rr * dt * dt
        V[0:3, 9:12] = V[0:3, 3:6]
        V[3:6, 3:6] = 0.5 * I3 * dt; V[3:6, 9:12] = 0.5 * I3 * dt
        V[6:9, 0:3] = 0.5 * Rd * dt
        V[6:9, 3:6] = 0.5 * -Rr @ Ra1 * dt * 0.5 * dt
        V[6:9, 6:9] = 0.5 * Rr * dt
        V[6:9, 9:12] = V[6:9, 3:6]
        V[9:12, 12:15] = I3 * dt; V[12:15, 15:18] = I3 * dt
        J = F @ J
        P = F @ P @ F.T + V @ N @ V.T
        dp, dv, dq = rp, rv, rq / np.linalg.norm(rq)
        sum_dt += dt
        a0, g0 = np.array(a1, float), np.array(g1, float)
    c = np.zeros(VIL_IMU_CONST)
    c[0:3], c[3:7], c[7:10], c[10:13], c[13:16], c[16] = dp, dq, dv, ba, bg, sum_dt
    c[17:26], c[26:35], c[35:44], c[44:53], c[53:62] = J[0:3, 9:12].ravel(), J[0:3, 12:15].ravel(), J[3:6, 12:15].ravel(), J[6:9, 9:12].ravel(), J[6:9, 12:15].ravel()
    c[62:287] = P.ravel()
    return c


def _qrot(q, v):
    u = q[:3]
    uv = 2 * np.cross(u, v)
    return v + q[3] * uv + np.cross(u, uv)


# ---- scene: truth + perturbed initial state for frames -1 .. K-1 ----------------------------------------
class Scene:
    def __init__(self, K, seed, phase=0.0):
        self.K, self.seed = K, seed
        self.rng = np.random.Generator(np.random.PCG64(seed))
        rng = self.rng
        self.traj = Trajectory(phase)
        self.nf = K + 1                      # frame f in [0, K] <-> time (f-1)*0.1 ; window = frames 1..K
        self.t = (np.arange(self.nf) - 1) * KF_DT
        self.ba_true = rng.normal(0, 0.02, 3); self.bg_true = rng.normal(0, 0.002, 3)
        self.pose_true = np.zeros((self.nf, 7)); self.sb_true = np.zeros((self.nf, 9))
        for f, t in enumerate(self.t):
            self.pose_true[f, :3] = self.traj.p(t); self.pose_true[f, 3:] = R_to_quat(self.traj.R(t))
            self.sb_true[f, :3] = self.traj.v(t); self.sb_true[f, 3:6] = self.ba_true; self.sb_true[f, 6:9] = self.bg_true
        self.ex_true = np.concatenate([TIC, R_to_quat(RIC)])
        # initial state: truth perturbed N(0,0.05 m), N(0,1 deg), N(0,0.1 m/s); biases ~ their priors
        self.pose_init = self.pose_true.copy(); self.sb_init = self.sb_true.copy()
        for f in range(self.nf):
            self.pose_init[f, :3] += rng.normal(0, 0.05, 3)
            dq = R_to_quat(expm_so3(rng.normal(0, np.deg2rad(1.0), 3)))
            q = qmul(self.pose_true[f, 3:], dq); self.pose_init[f, 3:] = q / np.linalg.norm(q)
            self.sb_init[f, :3] += rng.normal(0, 0.1, 3)
            self.sb_init[f, 3:6] = rng.normal(0, 0.02, 3); self.sb_init[f, 6:9] = rng.normal(0, 0.002, 3)
        self.ex_init = self.ex_true.copy()
        self.ex_init[:3] += rng.normal(0, 0.005, 3)
        q = qmul(self.ex_true[3:], R_to_quat(expm_so3(rng.normal(0, np.deg2rad(0.3), 3)))); self.ex_init[3:] = q / np.linalg.norm(q)
        self.td_true, self.td_init = 0.0, 3e-5   # yaml:113
        # IMU factors between consecutive frames, linearised at the initial bias estimate of frame f
        self.imu = np.zeros((self.nf - 1, VIL_IMU_CONST))
        for f in range(self.nf - 1):
            ts = self.t[f] + IMU_DT * np.arange(int(round(KF_DT / IMU_DT)) + 1)
            acc = np.array([self.traj.R(t).T @ (self.traj.a(t) + np.array([0, 0, G_NORM])) for t in ts]) + self.ba_true + rng.normal(0, ACC_N, (len(ts), 3))
            gyr = np.array([self.traj.w_body(t) for t in ts]) + self.bg_true + rng.normal(0, GYR_N, (len(ts), 3))
            self.imu[f] = preintegrate([IMU_DT] * (len(ts) - 1), acc[1:], gyr[1:], acc[0], gyr[0], self.sb_init[f, 3:6].copy(), self.sb_init[f, 6:9].copy())

    def cam_pose(self, t):
        R = self.traj.R(t); p = self.traj.p(t)
        return R @ RIC, R @ TIC + p  # R_wc, t_wc

    def project(self, Xw, t):
        Rwc, twc = self.cam_pose(t)
        Xc = Rwc.T @ (Xw - twc)
        return Xc

    # -- a window over scene frames [first, first+n)
    def window(self, first, n, L, n_plane=0, n_edge=0, n_icp=0, n_lps=0, anchor_only_first=False, lm_seed=0):
        rng = np.random.Generator(np.random.PCG64(self.seed * 1000 + 17 * first + lm_seed))
        w = Window(n, L)
        fr = np.arange(first, first + n)
        w.pose, w.speedbias = self.pose_init[fr].copy(), self.sb_init[fr].copy()
        w.ex_pose, w.td = self.ex_init.copy(), np.array([self.td_init])
        w.G = np.array([0, 0, G_NORM]); w.sqrt_info_px = FOCAL_LENGTH / 2.0; w.tr_over_row = 0.0
        w.q_lb, w.t_lb = R_to_quat(RLB), TLB.copy()
        w.imu_i, w.imu_j = np.arange(n - 1, dtype=np.int32), np.arange(1, n, dtype=np.int32)
        w.imu_const = self.imu[first:first + n - 1].copy()
        tt = self.t[fr]
        # landmarks
        vi, vj, vl, vc, lam_true = [], [], [], [], np.zeros(L)
        for l in range(L):
            if anchor_only_first:
                s, ln = 0, 2 + ((7 * l) % (n - 1))
            else:
                s = l % (n - 3)                              # start_frame < WINDOW_SIZE-2  (feature_manager.cpp:36)
                ln = 2 + ((7 * l) % (n - 1 - s))
            while True:
                depth = rng.uniform(2.0, 20.0); u, v = rng.uniform(-0.6, 0.6), rng.uniform(-0.45, 0.45)
                Rwc, twc = self.cam_pose(tt[s])
                Xw = Rwc @ (depth * np.array([u, v, 1.0])) + twc
                if all(self.project(Xw, tt[s + q])[2] > 0.5 for q in range(ln)):
                    break
            lam_true[l] = 1.0 / depth
            obs, vel = [], []
            for q in range(ln):
                t = tt[s + q]
                Xc = self.project(Xw, t)
                pt = np.array([Xc[0] / Xc[2], Xc[1] / Xc[2], 1.0])
                h = 5e-3
                Xa, Xb = self.project(Xw, t + h), self.project(Xw, t - h)
                vv = (Xa[:2] / Xa[2] - Xb[:2] / Xb[2]) / (2 * h)
                pt[:2] += rng.normal(0, 1.0 / FOCAL_LENGTH, 2)
                obs.append(pt); vel.append(vv)
            for q in range(1, ln):
                c = np.zeros(14)
                c[0:3], c[3:6], c[6:8], c[8:10] = obs[0], obs[q], vel[0], vel[q]
                c[10], c[11] = 0.0, 0.0                      # cur_td of both observations
                c[12], c[13] = obs[0][1] * FOCAL_LENGTH, obs[q][1] * FOCAL_LENGTH  # row - ROW/2
                vi.append(s); vj.append(s + q); vl.append(l); vc.append(c)
        w.vis_i, w.vis_j, w.vis_l = np.array(vi, np.int32), np.array(vj, np.int32), np.array(vl, np.int32)
        w.vis_const = np.array(vc).reshape(-1, 14)
        w.lm_const = (rng.uniform(size=L) < 0.3).astype(np.uint8)
        w.inv_depth = np.where(w.lm_const == 1, lam_true * (1 + rng.normal(0, 0.005, L)), lam_true * rng.uniform(0.8, 1.25, L))
        # LiDAR plane / edge points in a 20 x 20 x 5 m room
        lo, hi = np.array([-10.0, -10.0, -1.5]), np.array([10.0, 10.0, 3.5])
        if n_plane:
            k = np.arange(n_plane) * n // n_plane          # spread evenly, sorted by pose
            axis = rng.integers(0, 3, n_plane); side = rng.integers(0, 2, n_plane)
            P = rng.uniform(lo, hi, (n_plane, 3))
            P[np.arange(n_plane), axis] = np.where(side == 1, hi[axis], lo[axis])
            nrm = np.zeros((n_plane, 3)); nrm[np.arange(n_plane), axis] = np.where(side == 1, -1.0, 1.0)
            d_exact = -np.einsum("ij,ij->i", nrm, P)
            nrm_n = nrm + rng.normal(0, 0.01, (n_plane, 3)); nrm_n /= np.linalg.norm(nrm_n, axis=1, keepdims=True)
            d_n = d_exact + rng.normal(0, 0.005, n_plane)
            w.plane_pose = k.astype(np.int32)
            w.plane_const = np.concatenate([self._to_lidar(P, fr[k], rng), nrm_n, d_n[:, None]], axis=1)
        if n_edge:
            k = np.arange(n_edge) * n // n_edge
            ax = rng.integers(0, 3, n_edge)                 # edge direction axis
            C = rng.uniform(lo, hi, (n_edge, 3))
            for a in range(3):                              # the two other coordinates sit on the room boundary
                m = ax != a
                sidesel = rng.integers(0, 2, n_edge)
                C[m, a] = np.where(sidesel[m] == 1, hi[a], lo[a])
            dirv = np.zeros((n_edge, 3)); dirv[np.arange(n_edge), ax] = 1.0
            dirv += rng.normal(0, 0.01, (n_edge, 3)); dirv /= np.linalg.norm(dirv, axis=1, keepdims=True)
            A_, B_ = C + 0.1 * dirv, C - 0.1 * dirv          # localMapping.cpp:661-662
            w.edge_pose = k.astype(np.int32)
            w.edge_const = np.concatenate([self._to_lidar(C, fr[k], rng), A_, B_], axis=1)
        # ICP relative constraints (mode 3) and LPS rotation priors from the true trajectory
        if n_icp:
            ids, cc = [], []
            for q in range(n_icp):
                a = min(2 * q, n - 4)
                idq = [a, a + 1, a + 2, a + 3]
                ta, tb, tc, td_ = tt[idq]
                ti, tj = ta + 0.5 * (tb - ta), tc + 0.5 * (td_ - tc)
                Ri, Rj, pi, pj = self.traj.R(ti), self.traj.R(tj), self.traj.p(ti), self.traj.p(tj)
                pij = Ri.T @ (pj - pi) + rng.normal(0, 0.01, 3)
                ids.append(idq); cc.append([ta, tb, tc, td_, ti, tj, pij[0], pij[1], pij[2], 100.0 / 0.3])
            w.icp_ids, w.icp_const = np.array(ids, np.int32), np.array(cc)
        if n_lps:
            ids, cc = [], []
            for q in range(n_lps):
                l_ = min(1 + 2 * q, n - 2)
                tl, tr = tt[l_], tt[l_ + 1]
                tk = tl + 0.4 * (tr - tl)
                qk = R_to_quat(self.traj.R(tk) @ expm_so3(rng.normal(0, np.deg2rad(0.2), 3)))
                ids.append([l_, l_ + 1]); cc.append([tl, tr, tk, qk[0], qk[1], qk[2], qk[3]])
            w.lps_ids, w.lps_const = np.array(ids, np.int32), np.array(cc)
        w.truth = dict(pose=self.pose_true[fr].copy(), speedbias=self.sb_true[fr].copy(), ex_pose=self.ex_true.copy(), td=np.array([self.td_true]), inv_depth=lam_true)
        return w

    def _to_lidar(self, Pw, frames, rng):
        out = np.zeros_like(Pw)
        for f in np.unique(frames):
            m = frames == f
            R = quat_to_R(self.pose_true[f, 3:]); p = self.pose_true[f, :3]
            pb = (Pw[m] - p) @ R                            # R^T (pw - p)
            pl = pb @ RLB.T + TLB                            # p_l = RLB p_b + TLB
            rngs = np.linalg.norm(pl, axis=1, keepdims=True)
            pl = pl * (1 + rng.normal(0, 0.02, (m.sum(), 1)) / np.maximum(rngs, 1e-3))   # 2 cm range noise
            out[m] = pl
        return out


def make_config(config_id, prior_fn=None, **override):
    """Window for BASELINE.json config `config_id` (1..4).

    prior_fn(pre_window) -> abi.Prior | None builds the marginalisation prior from the preceding
    synthetic window (frames -1..K-1, landmarks anchored in frame -1); tests pass the oracle's
    marginalisation, bench.py the library's.  Without prior_fn a config that needs a prior gets
    `synthetic_prior`.
    """
    cfg = dict(CONFIGS[config_id]); cfg.update(override)
    seed = 20240601 + config_id + int(cfg.pop("seed_offset", 0))       # seed_offset: other scenes of the same shape (fuzz runs)
    sc = Scene(cfg["K"], seed)
    w = sc.window(1, cfg["K"], cfg["L"], cfg["n_plane"], cfg["n_edge"], cfg["n_icp"], cfg["n_lps"])
    w.config_id = config_id
    if cfg["prior"]:
        pre = sc.window(0, cfg["K"], max(30, cfg["L"] // 8), anchor_only_first=True, lm_seed=1)   # frames -1..K-2
        w.pre_window = pre
        pr = prior_fn(pre) if prior_fn is not None else None
        w.prior = pr if pr is not None else synthetic_prior(w, seed)
    return w


def synthetic_prior(w, seed):
    """Dense SPD prior around the initial state on poses 0..K-2, speedbias 0, ex, td (fallback only)."""
    rng = np.random.Generator(np.random.PCG64(seed + 99))
    K = w.K
    kinds = [BLK_POSE] * (K - 1) + [BLK_SPEEDBIAS, BLK_EX, BLK_TD]
    index = list(range(K - 1)) + [0, 0, 0]
    loc = [6] * (K - 1) + [9, 6, 1]
    cols = np.concatenate([[0], np.cumsum(loc)[:-1]]).astype(np.int32)
    n = int(sum(loc))
    sig = []
    for k, ls in zip(kinds, loc):
        sig += {BLK_POSE: [0.05] * 3 + [0.02] * 3, BLK_SPEEDBIAS: [0.1] * 3 + [0.02] * 3 + [0.002] * 3, BLK_EX: [0.01] * 3 + [0.005] * 3, BLK_TD: [0.001]}[k]
    Wd = np.diag(1.0 / np.array(sig))
    Q, _ = np.linalg.qr(rng.normal(size=(n, n)))
    M = np.eye(n) + 0.1 * rng.normal(size=(n, n)) / np.sqrt(n)
    J = Q @ Wd @ M                                          # dense, full rank
    x0 = np.concatenate([w.pose[:K - 1].ravel(), w.speedbias[0], w.ex_pose, w.td])
    r0 = rng.normal(0, 0.3, n)
    return Prior(n, kinds, index, cols, x0, J.T.ravel().copy(), r0)   # column-major flat

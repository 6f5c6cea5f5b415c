"""Synthetic replay (BASELINE.json configs[4]): the per-image loop around Estimator::optimization().

3indoor.bag (README.md:27-28) is not available offline, so config 5 runs on a SYNTHETIC REPLAY: a long
figure-eight sequence from the generator of synth.py, with the sliding window, marginalisation and the
feature bookkeeping active on every frame.  This module is harness-side host logic: it keeps what
`Estimator` / `FeatureManager` keep between two calls of optimization() and hands each window to a
backend (lib.Backend: the HIP library, or -- in tests / the CPU timing leg only -- the oracle):

  processImage      estimator.cpp:455-560   new frame: IMU-propagated initial state, observations appended
  optimization      estimator.cpp:1124-1687 window -> backend.solve, gauge fix (vil_gauge_fix), backend.marginalize
  setDepth/removeFailures feature_manager.cpp:150-168, 275-284
  slideWindow       estimator.cpp:1689-1790 MARGIN_OLD: drop frame 0 (removeBackShiftDepth, feature_manager.cpp:286-345)
                                            MARGIN_SECOND_NEW: drop frame K-2, merge its IMU samples (removeFront :369-388)
  window membership feature_manager.cpp:28-42,195-212: used_num >= 2 && start_frame < WINDOW_SIZE - 2
  ICP / LPS lists   estimator.cpp:1283-1286,1345-1348 (trimmed to 7 / 5), ids via stamp lookup (lidar_backend.cpp:3-93)

Keyframe selection (parallax test, feature_manager.cpp:46-92) is replaced by a fixed schedule: every
`second_new_every`-th frame is a non-keyframe.  Triangulation of new tracks is replaced by truth x U(0.8,1.25).
"""
import time

import numpy as np

from . import abi
from .abi import Window
from .synth import (Trajectory, preintegrate, quat_to_R, R_to_quat, qmul, expm_so3, RIC, TIC, RLB_RAW, TLB, _orthonormalise,
                    KF_DT, IMU_DT, G_NORM, FOCAL_LENGTH, ACC_N, GYR_N)

INIT_DEPTH = 5.0        # parameters.cpp:189


class Track:
    """FeaturePerId (feature_manager.h:50-78): observations of one landmark in consecutive window frames."""
    __slots__ = ("fid", "start", "obs", "depth", "lidar_flag", "future", "Xw", "slot")

    def __init__(self, fid, start, Xw, lidar_flag):
        self.fid, self.start, self.Xw, self.lidar_flag = fid, start, Xw, lidar_flag
        self.slot = -1           # track slot of the device-resident observation store (vil_win_*), handed out by Replay
        self.obs = []            # [(pt3, vel2, lidar_depth)] one per window frame start, start+1, ...
        self.depth = -1.0        # estimated_depth
        self.future = None       # {absolute frame: observation} still to arrive


class Replay:
    def __init__(self, K=10, n_frames=60, L=1000, n_plane=24000, n_edge=6000, seed=20240605, second_new_every=5, use_lidar_constraints=True,
                 max_iterations=8, max_time_s=0.0):
        self.K, self.NF = K, n_frames
        self.L_target, self.n_plane_pf, self.n_edge_pf = L, n_plane // K, n_edge // K
        self.second_new_every = second_new_every
        self.use_rel = use_lidar_constraints
        self.rng = rng = np.random.Generator(np.random.PCG64(seed))
        self.traj = Trajectory(0.0)
        self.RLB = _orthonormalise(RLB_RAW)
        self.t = np.arange(n_frames) * KF_DT
        self.ba_true, self.bg_true = rng.normal(0, 0.02, 3), rng.normal(0, 0.002, 3)
        self.pose_true = np.zeros((n_frames, 7)); self.v_true = np.zeros((n_frames, 3))
        for f, t in enumerate(self.t):
            self.pose_true[f, :3] = self.traj.p(t); self.pose_true[f, 3:] = R_to_quat(self.traj.R(t)); self.v_true[f] = self.traj.v(t)
        # raw IMU samples of interval f-1 -> f (index f), 200 Hz
        ns = int(round(KF_DT / IMU_DT))
        self.raw = [None]
        for f in range(1, n_frames):
            ts = self.t[f - 1] + IMU_DT * np.arange(ns + 1)
            acc = np.array([self.traj.R(t).T @ (self.traj.a(t) + np.array([0, 0, G_NORM])) for t in ts]) + self.ba_true + rng.normal(0, ACC_N, (ns + 1, 3))
            gyr = np.array([self.traj.w_body(t) for t in ts]) + self.bg_true + rng.normal(0, GYR_N, (ns + 1, 3))
            self.raw.append((acc, gyr))
        # The reference's configuration is max_num_iterations: 30 / max_solver_time: 0.05 s (config/mynteye_leishen_indoor.yaml:76-77 -> estimator.cpp:1404,1411):
        # bench.py's replay legs and the full-size replay test pass exactly that.  The DEFAULT here (8 iterations, no time cap) is the short cap of the small
        # parity chains in tests/ -- a labelled variant, not the yaml's value.  opts_parity: the same options without the wall-clock cap, for a CPU oracle that is
        # asked to reproduce the device's solve (a 50 ms cap cuts the CPU path after ~3 iterations of a configs[1]-sized window; the device never reaches it).
        self.opts = abi.default_options(max_iterations=max_iterations, max_time_s=max_time_s)
        self.opts_parity = abi.default_options(max_iterations=max_iterations)
        self._next_fid = 0
        self.max_tracks = 4096                               # track slots of the resident window (vil_win_cfg.max_tracks)
        self._free_slots = list(range(self.max_tracks - 1, -1, -1))
        self._spawn_rate = L / max(1.0, 1.75 * K - 4.5)      # new tracks per image; the divisor (measured) = mean number of windows a track is a member of
        self._init_window()

    # ---- bootstrapping: the first K frames stand in for the initial alignment (initialStructure) -----------------
    def _init_window(self):
        K, rng = self.K, self.rng
        self.frames = list(range(K))                         # absolute frame ids of the window
        self.pose = self.pose_true[:K].copy(); self.sb = np.zeros((K, 9))
        for k in range(K):
            self.pose[k, :3] += rng.normal(0, 0.05, 3)
            q = qmul(self.pose_true[k, 3:], R_to_quat(expm_so3(rng.normal(0, np.deg2rad(1.0), 3)))); self.pose[k, 3:] = q / np.linalg.norm(q)
            self.sb[k, :3] = self.v_true[k] + rng.normal(0, 0.1, 3)
            self.sb[k, 3:6] = rng.normal(0, 0.02, 3); self.sb[k, 6:9] = rng.normal(0, 0.002, 3)
        self.ex = np.concatenate([TIC, R_to_quat(RIC)])
        self.ex[:3] += rng.normal(0, 0.005, 3)
        q = qmul(self.ex[3:], R_to_quat(expm_so3(rng.normal(0, np.deg2rad(0.3), 3)))); self.ex[3:] = q / np.linalg.norm(q)
        self.td = np.array([3e-5])
        # per window slot k >= 1: raw samples + record of the interval (k-1 -> k)
        self.samples = [None] + [self._cat([self.raw[f]]) for f in range(1, K)]
        self.imu = [None] + [self._preint(self.samples[k], self.sb[k - 1]) for k in range(1, K)]
        self.tracks = []
        self.lidar = [self._lidar_points(f) for f in range(K)]
        self.icp, self.lps = [], []                          # (absolute frame ids..., constants)
        for f in range(K):
            self._spawn(f)
            self._observe(f, min(f, K - 1))
            self._rel_constraints(f)
        self.prior = abi.Prior()
        self.newest = K - 1

    @staticmethod
    def _cat(parts):
        acc = np.concatenate([p[0] if i == 0 else p[0][1:] for i, p in enumerate(parts)])
        gyr = np.concatenate([p[1] if i == 0 else p[1][1:] for i, p in enumerate(parts)])
        return acc, gyr

    def _preint(self, samples, sb_lin):
        acc, gyr = samples
        return preintegrate([IMU_DT] * (len(acc) - 1), acc[1:], gyr[1:], acc[0], gyr[0], sb_lin[3:6].copy(), sb_lin[6:9].copy())

    def _cam(self, f):
        R = quat_to_R(self.pose_true[f, 3:]); p = self.pose_true[f, :3]
        return R @ RIC, R @ TIC + p

    def _spawn(self, f):
        """New landmarks first seen in absolute frame f, with their whole future track precomputed."""
        rng, K = self.rng, self.K
        n_new = max(1, int(round(self._spawn_rate)))
        for _ in range(n_new):
            Rwc, twc = self._cam(f)
            depth = rng.uniform(2.0, 20.0); u, v = rng.uniform(-0.6, 0.6), rng.uniform(-0.45, 0.45)
            Xw = Rwc @ (depth * np.array([u, v, 1.0])) + twc
            ln = int(rng.integers(2, K + 3))
            flag = bool(rng.uniform() < 0.3)
            tr = Track(self._next_fid, -1, Xw, flag); self._next_fid += 1
            fut = {}
            for q in range(ln):
                g = f + q
                if g >= self.NF:
                    break
                Rc, tc = self._cam(g)
                Xc = Rc.T @ (Xw - tc)
                if Xc[2] < 0.5 or abs(Xc[0] / Xc[2]) > 0.9 or abs(Xc[1] / Xc[2]) > 0.7:
                    break
                h = 5e-3
                Xa = self._proj_t(Xw, self.t[g] + h); Xb = self._proj_t(Xw, self.t[g] - h)
                vel = (Xa[:2] / Xa[2] - Xb[:2] / Xb[2]) / (2 * h)
                pt = np.array([Xc[0] / Xc[2], Xc[1] / Xc[2], 1.0]); pt[:2] += rng.normal(0, 1.0 / FOCAL_LENGTH, 2)
                ld = Xc[2] * (1 + rng.normal(0, 0.005)) if flag else -1.0
                fut[g] = (pt, vel, ld)
            if len(fut) < 2:
                continue
            tr.future = fut
            tr.depth = depth * rng.uniform(0.8, 1.25)        # stands in for triangulate() (feature_manager.cpp:214-273)
            tr.slot = self._free_slots.pop()
            self.tracks.append(tr)

    def _proj_t(self, Xw, t):
        R = self.traj.R(t); p = self.traj.p(t)
        return (R @ RIC).T @ (Xw - (R @ TIC + p))

    def _observe(self, f, slot):
        """addFeatureCheckParallax (feature_manager.cpp:46-80): append the observations of image f at window slot `slot`."""
        for tr in self.tracks:
            ob = tr.future.pop(f, None)
            if ob is None:
                continue
            if not tr.obs:
                tr.start = slot
                if ob[2] > 0:
                    tr.depth = ob[2]                         # LiDAR depth of the anchor observation
            tr.obs.append(ob)
        # tracks that lost their feature before ever entering a window and have nothing left are dropped lazily in _window()

    def _lidar_points(self, f):
        """Plane / edge correspondences of the scan attached to absolute frame f (synth.Scene.window recipe)."""
        rng = self.rng
        lo, hi = np.array([-10.0, -10.0, -1.5]), np.array([10.0, 10.0, 3.5])
        R = quat_to_R(self.pose_true[f, 3:]); p = self.pose_true[f, :3]

        def to_lidar(Pw):
            pl = ((Pw - p) @ R) @ self.RLB.T + TLB
            rn = np.linalg.norm(pl, axis=1, keepdims=True)
            return pl * (1 + rng.normal(0, 0.02, (len(pl), 1)) / np.maximum(rn, 1e-3))
        npl, ne = self.n_plane_pf, self.n_edge_pf
        plane = np.zeros((0, 7)); edge = np.zeros((0, 9))
        if npl:
            axis = rng.integers(0, 3, npl); side = rng.integers(0, 2, npl)
            P = rng.uniform(lo, hi, (npl, 3)); P[np.arange(npl), axis] = np.where(side == 1, hi[axis], lo[axis])
            nrm = np.zeros((npl, 3)); nrm[np.arange(npl), axis] = np.where(side == 1, -1.0, 1.0)
            d = -np.einsum("ij,ij->i", nrm, P) + rng.normal(0, 0.005, npl)
            nrm = nrm + rng.normal(0, 0.01, (npl, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
            plane = np.concatenate([to_lidar(P), nrm, d[:, None]], axis=1)
        if ne:
            ax = rng.integers(0, 3, ne); Cp = rng.uniform(lo, hi, (ne, 3))
            for a in range(3):
                m = ax != a; sel = rng.integers(0, 2, ne)
                Cp[m, a] = np.where(sel[m] == 1, hi[a], lo[a])
            dv = np.zeros((ne, 3)); dv[np.arange(ne), ax] = 1.0
            dv += rng.normal(0, 0.01, (ne, 3)); dv /= np.linalg.norm(dv, axis=1, keepdims=True)
            edge = np.concatenate([to_lidar(Cp), Cp + 0.1 * dv, Cp - 0.1 * dv], axis=1)
        return plane, edge

    def _rel_constraints(self, f):
        """Scan-to-scan ICP constraint every second frame (brackets f-3,f-2 | f-1,f) and an LPS rotation prior per frame."""
        if not self.use_rel:
            return
        rng = self.rng
        if f >= 3 and f % 2 == 1:
            ids = (f - 3, f - 2, f - 1, f)
            ta, tb, tc, td_ = self.t[list(ids)]
            ti, tj = ta + 0.5 * (tb - ta), tc + 0.5 * (td_ - tc)
            Ri, pi, pj = self.traj.R(ti), self.traj.p(ti), self.traj.p(tj)
            pij = Ri.T @ (pj - pi) + rng.normal(0, 0.01, 3)
            self.icp.append((ids, np.array([ta, tb, tc, td_, ti, tj, pij[0], pij[1], pij[2], 100.0 / 0.3])))
            self.icp = self.icp[-5:]                          # estimator.cpp:1345-1348
        if f >= 1:
            tl, tr = self.t[f - 1], self.t[f]
            tk = tl + 0.4 * (tr - tl)
            qk = R_to_quat(self.traj.R(tk) @ expm_so3(rng.normal(0, np.deg2rad(0.2), 3)))
            self.lps.append(((f - 1, f), np.array([tl, tr, tk, qk[0], qk[1], qk[2], qk[3]])))
            self.lps = self.lps[-7:]                          # estimator.cpp:1283-1286

    # ---- one window for the backend ----------------------------------------------------------------------------------
    def window(self, with_lidar=True):
        """with_lidar=False: the point factors are resident on the device as frame slabs (lib.Backend.lidar_push of self.lidar[k]);
        the window carries no LiDAR tables and asks for them with n_plane = n_edge = VIL_LIDAR_RESIDENT."""
        K = self.K
        sel = [tr for tr in self.tracks if len(tr.obs) >= 2 and tr.start < K - 3]          # feature_manager.cpp:36
        w = Window(K, len(sel))
        w.pose, w.speedbias, w.ex_pose, w.td = self.pose.copy(), self.sb.copy(), self.ex.copy(), self.td.copy()
        w.G = np.array([0, 0, G_NORM]); w.sqrt_info_px = FOCAL_LENGTH / 2.0; w.tr_over_row = 0.0
        w.q_lb, w.t_lb = R_to_quat(self.RLB), TLB.copy()
        w.imu_i, w.imu_j = np.arange(K - 1, dtype=np.int32), np.arange(1, K, dtype=np.int32)
        w.imu_const = np.array(self.imu[1:])
        vi, vj, vl, vc = [], [], [], []
        lam = np.zeros(len(sel)); lmc = np.zeros(len(sel), np.uint8)
        for l, tr in enumerate(sel):
            lam[l] = 1.0 / (tr.depth if tr.depth > 0 else INIT_DEPTH)                      # feature_manager.cpp:206-209
            lmc[l] = 1 if tr.lidar_flag else 0
            p0, v0, _ = tr.obs[0]
            for q in range(1, len(tr.obs)):
                pq, vq, _ = tr.obs[q]
                vi.append(tr.start); vj.append(tr.start + q); vl.append(l)
                vc.append((p0[0], p0[1], p0[2], pq[0], pq[1], pq[2], v0[0], v0[1], vq[0], vq[1], 0.0, 0.0, p0[1] * FOCAL_LENGTH, pq[1] * FOCAL_LENGTH))
        w.vis_i, w.vis_j, w.vis_l = np.array(vi, np.int32), np.array(vj, np.int32), np.array(vl, np.int32)
        w.vis_const = np.array(vc).reshape(-1, 14)
        w.inv_depth, w.lm_const = lam, lmc
        if with_lidar:
            w.plane_pose = np.concatenate([np.full(len(self.lidar[k][0]), k, np.int32) for k in range(K)])
            w.plane_const = np.concatenate([self.lidar[k][0] for k in range(K)])
            w.edge_pose = np.concatenate([np.full(len(self.lidar[k][1]), k, np.int32) for k in range(K)])
            w.edge_const = np.concatenate([self.lidar[k][1] for k in range(K)])
        else:
            w.lidar_resident = True
        slot = {f: k for k, f in enumerate(self.frames)}
        icp_ids, icp_c, lps_ids, lps_c = [], [], [], []
        icp_marg = lps_marg = -1
        for ids, c in self.icp:                               # FindWindowsID: every stamp must be a window frame
            if all(i in slot for i in ids):
                a, b, c_, d = (slot[i] for i in ids)
                if b > a and d > c_ and a != c_:
                    if a == 0:
                        icp_marg = len(icp_ids)               # estimator.cpp:1381-1389 (the last one wins)
                    icp_ids.append((a, b, c_, d)); icp_c.append(c)
        stamps = self.t[self.frames]
        for ids, c in self.lps:                               # FindNearest2ID + bracket gap < 0.2 s (estimator.cpp:1307-1323)
            lb = int(np.searchsorted(stamps, c[2], side="left"))
            if lb > K - 1 or lb - 1 < 0:
                continue
            il, ir = lb - 1, lb
            if stamps[ir] - stamps[il] < 0.2:
                cc = c.copy(); cc[0], cc[1] = stamps[il], stamps[ir]
                if il == 0:
                    lps_marg = len(lps_ids)
                lps_ids.append((il, ir)); lps_c.append(cc)
        if icp_ids:
            w.icp_ids, w.icp_const = np.array(icp_ids, np.int32), np.array(icp_c)
        if lps_ids:
            w.lps_ids, w.lps_const = np.array(lps_ids, np.int32), np.array(lps_c)
        w.prior = self.prior
        w._sel, w._icp_marg, w._lps_marg = sel, icp_marg, lps_marg
        return w

    def margin_flag(self):
        f = self.newest
        return abi.MARGIN_SECOND_NEW if (self.second_new_every and f % self.second_new_every == self.second_new_every - 1) else abi.MARGIN_OLD

    # ---- after the backend: double2vector bookkeeping + slideWindow --------------------------------------------------
    def absorb(self, w, prior_out, flag):
        """Write the solved (gauge-fixed) window back, adopt the new prior, slide, and take in the next image."""
        K = self.K
        self.pose, self.sb, self.ex, self.td = w.pose.copy(), w.speedbias.copy(), w.ex_pose.copy(), w.td.copy()
        dead = set()
        for l, tr in enumerate(w._sel):                       # setDepth + removeFailures
            tr.depth = 1.0 / w.inv_depth[l]
            if tr.depth < 0:
                dead.add(tr.fid)
        if dead:
            self.tracks = [tr for tr in self.tracks if tr.fid not in dead]
        if prior_out is not None:
            pr = prior_out.to_prior()
            if pr is not None:
                self.prior = pr
        if flag == abi.MARGIN_OLD:
            R0 = quat_to_R(self.pose[0, 3:]) @ quat_to_R(self.ex[3:]); P0 = self.pose[0, :3] + quat_to_R(self.pose[0, 3:]) @ self.ex[:3]
            R1 = quat_to_R(self.pose[1, 3:]) @ quat_to_R(self.ex[3:]); P1 = self.pose[1, :3] + quat_to_R(self.pose[1, 3:]) @ self.ex[:3]
            keep = []
            for tr in self.tracks:                            # removeBackShiftDepth
                if not tr.obs:
                    keep.append(tr); continue
                if tr.start != 0:
                    tr.start -= 1; keep.append(tr); continue
                uv, _, ld = tr.obs[0]
                depth = ld if ld > 0 else (tr.depth if tr.depth > 0 else -1.0)
                tr.obs.pop(0)
                if len(tr.obs) < 2:
                    continue
                pj = R1.T @ (R0 @ (uv * depth) + P0 - P1)
                if tr.obs[0][2] > 0:
                    tr.depth, tr.lidar_flag = tr.obs[0][2], True
                elif pj[2] > 0:
                    tr.depth, tr.lidar_flag = pj[2], False
                else:
                    tr.depth, tr.lidar_flag = INIT_DEPTH, False
                keep.append(tr)
            self.tracks = keep
            self.frames = self.frames[1:]
            self.pose = np.vstack([self.pose[1:], self.pose[-1:]]); self.sb = np.vstack([self.sb[1:], self.sb[-1:]])
            self.samples = [None] + self.samples[2:]; self.imu = [None] + self.imu[2:]
            self.lidar = self.lidar[1:]
        else:
            keep = []
            for tr in self.tracks:                            # removeFront(frame_count = K-1)
                if not tr.obs:
                    keep.append(tr); continue
                if tr.start == K - 1:
                    tr.start -= 1; keep.append(tr); continue
                end = tr.start + len(tr.obs) - 1
                if end < K - 2:
                    keep.append(tr); continue
                tr.obs.pop(K - 2 - tr.start)
                if tr.obs or tr.future:
                    keep.append(tr)
            self.tracks = keep
            self.frames = self.frames[:K - 2] + [self.frames[K - 1]]
            self.pose = np.vstack([self.pose[:K - 2], self.pose[K - 1:], self.pose[K - 1:]]); self.sb = np.vstack([self.sb[:K - 2], self.sb[K - 1:], self.sb[K - 1:]])
            merged = self._cat([self.samples[K - 2], self.samples[K - 1]])                 # push_back of the newest interval's samples (:1763-1772)
            lin = np.concatenate([np.zeros(3), self.imu[K - 2][10:13], self.imu[K - 2][13:16]])
            self.samples = self.samples[:K - 2] + [merged]; self.imu = self.imu[:K - 2] + [self._preint(merged, lin)]
            self.lidar = self.lidar[:K - 2] + [self.lidar[K - 1]]
        self.tracks = [tr for tr in self.tracks if tr.obs or tr.future]
        used = {tr.slot for tr in self.tracks}               # slots of tracks that are gone can be handed out again
        self._free_slots = [q for q in range(self.max_tracks - 1, -1, -1) if q not in used]
        # next image
        self.newest += 1
        f = self.newest
        if f >= self.NF:
            return False
        self.frames.append(f)
        self.samples.append(self._cat([self.raw[f]]))
        rec = self._preint(self.samples[K - 1], self.sb[K - 2])
        self.imu.append(rec)
        # processIMU (estimator.cpp:170-200) propagates the newest state through the samples; first order: use the pre-integrated deltas
        Ri = quat_to_R(self.pose[K - 2, 3:]); dt = rec[16]; g = np.array([0, 0, G_NORM])
        self.pose[K - 1, :3] = self.pose[K - 2, :3] + self.sb[K - 2, :3] * dt - 0.5 * g * dt * dt + Ri @ rec[0:3]
        q = qmul(self.pose[K - 2, 3:], rec[3:7]); self.pose[K - 1, 3:] = q / np.linalg.norm(q)
        self.sb[K - 1, :3] = self.sb[K - 2, :3] - g * dt + Ri @ rec[7:10]
        self.sb[K - 1, 3:] = self.sb[K - 2, 3:]
        self.lidar.append(self._lidar_points(f))
        self._spawn(f)
        self._observe(f, K - 1)
        self._rel_constraints(f)
        return True

    # ---- the fully resident window (vil_win_*): what crosses PCIe per image ---------------------------------------------
    def win_open_args(self):
        from .synth import ACC_W, GYR_W
        return dict(K=self.K, max_tracks=self.max_tracks, max_samples=256, noise=(ACC_N, GYR_N, ACC_W, GYR_W), G=(0.0, 0.0, G_NORM),
                    sqrt_info_px=FOCAL_LENGTH / 2.0, tr_over_row=0.0, q_lb=R_to_quat(self.RLB), t_lb=TLB, use_td=1)

    def win_frame(self, k):
        """vil_win_frame of window frame k: the IMU samples of the interval ending in it (its first measurement and the bias
        linearisation point of frame k-1), the observations made in it (by track slot), its LiDAR points."""
        fr = dict(dt=np.zeros(0), acc=np.zeros((0, 3)), gyr=np.zeros((0, 3)), acc0=np.zeros(3), gyr0=np.zeros(3), lin_ba=np.zeros(3), lin_bg=np.zeros(3))
        if k >= 1:
            acc, gyr = self.samples[k]
            fr.update(dt=np.full(len(acc) - 1, IMU_DT), acc=acc[1:], gyr=gyr[1:], acc0=acc[0], gyr0=gyr[0], lin_ba=self.imu[k][10:13], lin_bg=self.imu[k][13:16])
        trk, obs = [], []
        for tr in self.tracks:
            q = k - tr.start
            if tr.obs and 0 <= q < len(tr.obs):
                pt, vel, _ = tr.obs[q]
                trk.append(tr.slot); obs.append((pt[0], pt[1], pt[2], vel[0], vel[1], 0.0, pt[1] * FOCAL_LENGTH, 0.0))
        fr.update(obs_track=np.array(trk, np.int32), obs=np.array(obs).reshape(-1, 8), plane=self.lidar[k][0], edge=self.lidar[k][1])
        return fr

    def win_window(self):
        """The window for vil_win_solve: the small tables only -- visual structure per landmark, no factor constants, no LiDAR, no IMU records, no prior."""
        w = self.window(with_lidar=False)
        w._fix()                                             # (dtypes / contiguity settled here: Backend.win_solve takes the arrays as they are)
        w.lm_track = np.array([tr.slot for tr in w._sel], np.int32)
        w.lm_start = np.array([tr.start for tr in w._sel], np.int32)
        w.lm_nobs = np.array([len(tr.obs) for tr in w._sel], np.int32)
        return w

    def truth_window(self):
        return self.pose_true[self.frames]


def run(backend, rp, n_steps=None, on_frame=None, with_marg=True, log_path=None):
    """Drive `rp` with `backend` for n_steps images; returns per-frame records (latencies in ms, errors vs truth).
    log_path: append the newest pose of every image in the reference's Frontend.txt format (visualization.cpp:199-212)."""
    out = []
    log = open(log_path, "a") if log_path else None
    step = 0
    while n_steps is None or step < n_steps:
        w = rp.window()
        flag = rp.margin_flag()
        p0 = w.pose[0].copy()
        t0 = time.perf_counter()
        summ = backend.solve(w, rp.opts)
        t1 = time.perf_counter()
        backend.gauge_fix(p0, w)
        t2 = time.perf_counter()
        po = backend.marginalize(w, flag, w._icp_marg, w._lps_marg, rp.opts) if with_marg else None
        t3 = time.perf_counter()
        tw = rp.truth_window()
        rec = dict(frame=rp.newest, L=w.L, n_vis=len(w.vis_i), n_lidar=len(w.plane_pose) + len(w.edge_pose), n_icp=len(w.icp_ids), n_lps=len(w.lps_ids),
                   prior_n=w.prior.n, flag=int(flag), iterations=summ.iterations, termination=int(summ.termination), initial_cost=summ.initial_cost,
                   final_cost=summ.final_cost, solve_ms=1e3 * (t1 - t0), marg_ms=1e3 * (t3 - t2),
                   pos_err_newest=float(np.linalg.norm(w.pose[-1, :3] - tw[-1, :3])), new_prior_n=(po.c.n if po is not None else -1))
        if on_frame is not None:
            on_frame(rp, w, po, rec)
        if log is not None:
            from . import formats
            log.write(formats.format_trajectory_line(float(rp.t[rp.newest]), w.pose[-1, :3], w.pose[-1, 3:7]))
        out.append(rec)
        step += 1
        if not rp.absorb(w, po, flag):
            break
    if log is not None:
        log.close()
    return out

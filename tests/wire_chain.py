"""Test infrastructure for examples/wire_replay.cpp (end to end): a synthetic RECORDED SEQUENCE of what reaches the estimator node --
per image the /feature_tracker_/feature message (float32 points + 6 float32 channels, feature_tracker_node.cpp:127-177), the IMU samples of the
interval, the LiDAR correspondences -- and a Python mirror of the per-image loop around Estimator::optimization() that any backend (the HIP library
through its CLASSIC entry points, or the CPU oracle) can be driven with:

   decode (formats.decode_feature_cloud)  ->  RefFeatureManager (the line-by-line transcription of feature_manager.cpp in
   test_feature_table_hypothesis.py)  ->  triangulate  ->  abi.Window with every table on the host  ->  backend.solve / gauge_fix / marginalize
   ->  setDepth / slideWindow / removeFailures  ->  Frontend.txt line (formats.format_trajectory_line)

The C++ example walks the same sequence through vil::FeatureTable / TrackSlots / WindowFrames and the fully resident window (vil_win_*); the test
compares the three trajectory logs.  3indoor.bag is not available offline: the sequence is synthetic (synth.py / replay.py generators)."""
import struct

import numpy as np

from mvil_fusion_amd import abi, formats, replay, synth
from mvil_fusion_amd.abi import Window

import test_feature_table_hypothesis as tfm

FOCAL, COL_HALF, ROW_HALF = 460.0, 376.0, 240.0
INIT_DEPTH, MIN_PARALLAX = 5.0, 10.0 / 460.0


def make_sequence(K=8, n_images=24, new_per_image=26, n_plane=1600, n_edge=480, seed=5, max_iterations=8):
    """-> dict(header..., frames=[record per image]); frames 0 .. K-1 carry the initial-alignment stand-in state."""
    NF = K + n_images
    rp = replay.Replay(K=K, n_frames=NF + 2, L=100, n_plane=n_plane, n_edge=n_edge, seed=seed, use_lidar_constraints=False, max_iterations=max_iterations)
    rng = np.random.default_rng(seed + 77)
    # landmark tracks: born in image f, seen for 2 .. K + 2 consecutive images while they stay in the field of view
    obs_of = [[] for _ in range(NF)]
    fid = 0
    for f in range(NF):
        for _ in range(new_per_image):
            Rwc, twc = rp._cam(f)
            depth = rng.uniform(2.0, 20.0); u, v = rng.uniform(-0.6, 0.6), rng.uniform(-0.45, 0.45)
            Xw = Rwc @ (depth * np.array([u, v, 1.0])) + twc
            ln = int(rng.integers(2, K + 3)); lidar = bool(rng.uniform() < 0.3)
            for q in range(ln):
                g = f + q
                if g >= NF:
                    break
                Rc, tc = rp._cam(g)
                Xc = Rc.T @ (Xw - tc)
                if Xc[2] < 0.5 or abs(Xc[0] / Xc[2]) > 0.9 or abs(Xc[1] / Xc[2]) > 0.7:
                    break
                h = 5e-3
                Xa = rp._proj_t(Xw, rp.t[g] + h); Xb = rp._proj_t(Xw, rp.t[g] - h)
                vel = (Xa[:2] / Xa[2] - Xb[:2] / Xb[2]) / (2 * h)
                x, y = Xc[0] / Xc[2] + rng.normal(0, 1.0 / FOCAL), Xc[1] / Xc[2] + rng.normal(0, 1.0 / FOCAL)
                ld = Xc[2] * (1 + rng.normal(0, 0.005)) if lidar else -1.0
                obs_of[g].append((fid, x, y, 1.0, x * FOCAL + COL_HALF, y * FOCAL + ROW_HALF, vel[0], vel[1], ld))
            fid += 1
    frames = []
    for f in range(NF):
        o = obs_of[f]
        perm = rng.permutation(len(o))                                       # the tracker publishes in its own order; the estimator sorts by id (std::map)
        ids = np.array([o[q][0] for q in perm], np.int64); obs8 = np.array([o[q][1:] for q in perm], np.float64).reshape(-1, 8)
        pts, ch = formats.encode_feature_cloud(ids, np.zeros(len(ids), np.int64), obs8, num_of_cam=1)
        if f >= 1:
            acc, gyr = rp.raw[f]
            dt, a, g_, first = np.full(len(acc) - 1, synth.IMU_DT), acc[1:], gyr[1:], np.concatenate([acc[0], gyr[0]])
        else:
            dt, a, g_, first = np.zeros(0), np.zeros((0, 3)), np.zeros((0, 3)), np.zeros(6)
        plane, edge = rp.lidar[f] if f < K else rp._lidar_points(f)
        init = np.concatenate([rp.pose[f], rp.sb[f]]) if f < K else np.zeros(0)
        frames.append(dict(stamp=float(rp.t[f]), dt=dt, acc=a, gyr=g_, first=first, points=pts, channels=ch, plane=plane, edge=edge, init=init))
    o = rp.win_open_args()
    return dict(K=K, n_images=n_images, max_tracks=2048, max_samples=256, use_td=1, row_half=int(ROW_HALF), num_of_cam=1, noise=o["noise"], G=o["G"], sqrt_info_px=o["sqrt_info_px"],
                tr_over_row=0.0, q_lb=o["q_lb"], t_lb=o["t_lb"], init_depth=INIT_DEPTH, min_parallax=MIN_PARALLAX, td=float(rp.td[0]), ex=rp.ex.copy(), max_iterations=max_iterations,
                frames=frames, truth=rp.pose_true.copy())


def write_sequence(path, seq):
    def arr(f, a, dt):
        a = np.ascontiguousarray(np.asarray(a, dt)).ravel()
        f.write(struct.pack("<q", a.size)); f.write(a.tobytes())
    with open(path, "wb") as f:
        arr(f, [seq["K"], seq["max_tracks"], seq["max_samples"], seq["use_td"], seq["n_images"], seq["row_half"], seq["num_of_cam"]], np.int32)
        arr(f, list(seq["noise"]) + list(seq["G"]) + [seq["sqrt_info_px"], seq["tr_over_row"]] + list(seq["q_lb"]) + list(seq["t_lb"]) +
            [seq["init_depth"], seq["min_parallax"], seq["td"]] + list(seq["ex"]) + [float(seq["max_iterations"])], np.float64)
        for fr in seq["frames"]:
            arr(f, [fr["stamp"]], np.float64); arr(f, fr["dt"], np.float64); arr(f, fr["acc"], np.float64); arr(f, fr["gyr"], np.float64); arr(f, fr["first"], np.float64)
            arr(f, fr["points"], np.float32)
            for c in fr["channels"]:
                arr(f, c, np.float32)
            arr(f, fr["plane"], np.float64); arr(f, fr["edge"], np.float64); arr(f, fr["init"], np.float64)


# ---- Python mirror of the per-image loop (test infrastructure) -------------------------------------------------------------------------------------
def _cam_pose(pose7, ex):
    R = synth.quat_to_R(pose7[3:]); Ric = synth.quat_to_R(ex[3:])
    return R @ Ric, pose7[:3] + R @ ex[:3]


def _triangulate(fm, pose, ex, W):
    """feature_manager.cpp:214-273 on the transcription's tracks (JacobiSVD -> numpy SVD)."""
    for it in fm.feature:
        if not (len(it["fpf"]) >= 2 and it["start_frame"] < W - 2) or it["estimated_depth"] > 0:
            continue
        R0, t0 = _cam_pose(pose[it["start_frame"]], ex)
        rows = []
        for m, fpf in enumerate(it["fpf"]):
            R1, t1 = _cam_pose(pose[it["start_frame"] + m], ex)
            t = R0.T @ (t1 - t0); R = R0.T @ R1
            P = np.hstack([R.T, (-R.T @ t)[:, None]])
            f = fpf["point"] / np.linalg.norm(fpf["point"])
            rows.append(f[0] * P[2] - f[2] * P[0]); rows.append(f[1] * P[2] - f[2] * P[1])
        V = np.linalg.svd(np.array(rows))[2][-1]
        d = V[2] / V[3]
        it["estimated_depth"] = INIT_DEPTH if d < 0 else d


def run_chain(backend, seq, log_path=None, shadow=None):
    """The estimator's per-image loop in Python on the CLASSIC entry points of `backend` (lib.Backend: HIP library or oracle).  Returns the Frontend.txt
    text and per-image records.  shadow: a second backend that is handed a copy of EVERY image's input window (same tables, same prior, same state) and
    whose results are logged beside the chain's but never fed back -- the comparator of a chain must not run its own: a landmark whose solved inverse
    depth sits at zero is removed or kept on the sign of rounding noise (setDepth / removeFailures, feature_manager.cpp:150-179), and two independent
    chains then differ by a landmark from that image on.  With a shadow the return value is (text, records, shadow text, shadow records)."""
    K = seq["K"]; W = K - 1
    tfm.W, tfm.INIT_DEPTH, tfm.MIN_PARALLAX = W, seq["init_depth"], seq["min_parallax"]          # the transcription reads its constants from the module
    fm = tfm.RefFeatureManager()
    pose = np.zeros((K, 7)); sb = np.zeros((K, 9)); stamp = np.zeros(K)
    ex, td = seq["ex"].copy(), np.array([seq["td"]])
    samples = [None] * K           # per window frame: (dt, acc, gyr, acc0, gyr0, lin_ba, lin_bg) of the interval that ends in it
    lidar = [None] * K
    G = np.array(seq["G"])
    opts = abi.default_options(max_iterations=seq["max_iterations"])
    prior = abi.Prior()
    state = dict(kf=True)

    def take(k, fr, bootstrap):
        if bootstrap:
            pose[k], sb[k] = fr["init"][:7], fr["init"][7:]
        stamp[k] = fr["stamp"]
        if bootstrap or samples[k] is None:
            samples[k] = [np.zeros(0), np.zeros((0, 3)), np.zeros((0, 3)), fr["first"][:3].copy(), fr["first"][3:].copy(), sb[k, 3:6].copy(), sb[k, 6:9].copy()]
        s = samples[k]
        a0, g0 = s[3].copy(), s[4].copy()
        if len(s[0]):                                                      # (after MARGIN_SECOND_NEW nothing is left here: the samples were merged into k - 1)
            raise AssertionError("interval not reset")
        for dt, a1, g1 in zip(fr["dt"], fr["acc"], fr["gyr"]):
            if not bootstrap:                                              # processIMU, estimator.cpp:109-116
                R = synth.quat_to_R(pose[k, 3:])
                ua0 = R @ (a0 - sb[k, 3:6]) - G
                th = (0.5 * (g0 + g1) - sb[k, 6:9]) * dt
                dq = np.array([0.5 * th[0], 0.5 * th[1], 0.5 * th[2], 1.0]); dq /= np.linalg.norm(dq)
                q = synth.qmul(pose[k, 3:], dq); pose[k, 3:] = q / np.linalg.norm(q)
                R = synth.quat_to_R(pose[k, 3:])
                ua1 = R @ (a1 - sb[k, 3:6]) - G
                ua = 0.5 * (ua0 + ua1)
                pose[k, :3] += dt * sb[k, :3] + 0.5 * dt * dt * ua
                sb[k, :3] += dt * ua
            a0, g0 = a1, g1
        s[0], s[1], s[2] = np.asarray(fr["dt"], float), np.asarray(fr["acc"], float).reshape(-1, 3), np.asarray(fr["gyr"], float).reshape(-1, 3)
        lidar[k] = (fr["plane"], fr["edge"])
        ids, cams, obs8 = formats.decode_feature_cloud(fr["points"], fr["channels"], num_of_cam=seq["num_of_cam"])
        image = {int(i): [(int(c), list(o))] for i, c, o in zip(ids, cams, obs8)}
        state["kf"] = fm.addFeatureCheckParallax(k, image, float(td[0]))

    def reset_interval(k, first):
        samples[k] = [np.zeros(0), np.zeros((0, 3)), np.zeros((0, 3)), first[:3].copy(), first[3:].copy(), sb[k, 3:6].copy(), sb[k, 6:9].copy()]

    frames = seq["frames"]
    for k in range(K):
        take(k, frames[k], True)
    lines, recs, slines, srecs = [], [], [], []
    for img in range(seq["n_images"]):
        flag = abi.MARGIN_OLD if state["kf"] else abi.MARGIN_SECOND_NEW
        _triangulate(fm, pose, ex, W)
        sel = [it for it in fm.feature if len(it["fpf"]) >= 2 and it["start_frame"] < W - 2]
        w = Window(K, len(sel))
        w.pose, w.speedbias, w.ex_pose, w.td = pose.copy(), sb.copy(), ex.copy(), td.copy()
        w.G = G.copy(); w.sqrt_info_px = seq["sqrt_info_px"]; w.tr_over_row = seq["tr_over_row"]
        w.q_lb, w.t_lb = np.array(seq["q_lb"]), np.array(seq["t_lb"])
        w.imu_i, w.imu_j = np.arange(K - 1, dtype=np.int32), np.arange(1, K, dtype=np.int32)
        w.imu_const = np.array([synth.preintegrate(list(samples[k][0]), samples[k][1], samples[k][2], samples[k][3], samples[k][4], samples[k][5].copy(), samples[k][6].copy()) for k in range(1, K)])
        vi, vj, vl, vc = [], [], [], []
        for l, it in enumerate(sel):
            f0 = it["fpf"][0]
            for m in range(1, len(it["fpf"])):
                fj = it["fpf"][m]
                vi.append(it["start_frame"]); vj.append(it["start_frame"] + m); vl.append(l)
                vc.append(list(f0["point"]) + list(fj["point"]) + list(f0["velocity"]) + list(fj["velocity"]) + [f0["cur_td"], fj["cur_td"], f0["uv"][1] - seq["row_half"], fj["uv"][1] - seq["row_half"]])
        w.vis_i, w.vis_j, w.vis_l = np.array(vi, np.int32), np.array(vj, np.int32), np.array(vl, np.int32)
        w.vis_const = np.array(vc).reshape(-1, 14)
        w.inv_depth = np.array(fm.getDepthVector(), float).reshape(-1); w.lm_const = np.array([1 if it["lidar_depth_flag"] else 0 for it in sel], np.uint8)
        w.plane_pose = np.concatenate([np.full(len(lidar[k][0]), k, np.int32) for k in range(K)]); w.plane_const = np.concatenate([lidar[k][0] for k in range(K)])
        w.edge_pose = np.concatenate([np.full(len(lidar[k][1]), k, np.int32) for k in range(K)]); w.edge_const = np.concatenate([lidar[k][1] for k in range(K)])
        w.prior = prior
        p0 = w.pose[0].copy()
        if shadow is not None:
            ws = Window.from_dict(w.to_dict()); ws.prior = prior
            ss = shadow.solve(ws, opts); shadow.gauge_fix(p0, ws)
            ps = shadow.marginalize(ws, flag, -1, -1, opts)
            slines.append(formats.format_trajectory_line(stamp[K - 1], ws.pose[K - 1, :3], ws.pose[K - 1, 3:]))
            srecs.append(dict(flag=int(flag), L=len(sel), iterations=ss.iterations, final_cost=ss.final_cost, n=ps.c.n))
        summ = backend.solve(w, opts); backend.gauge_fix(p0, w)
        pose[:], sb[:], ex[:], td[:] = w.pose, w.speedbias, w.ex_pose, w.td
        fm.setDepth(list(w.inv_depth))
        po = backend.marginalize(w, flag, -1, -1, opts)
        prior = po.to_prior() or prior
        lines.append(formats.format_trajectory_line(stamp[K - 1], pose[K - 1, :3], pose[K - 1, 3:]))
        recs.append(dict(flag=int(flag), L=len(sel), iterations=summ.iterations, final_cost=summ.final_cost, n=po.c.n))
        if img + 1 == seq["n_images"]:
            break
        nxt = frames[K + img]
        if flag == abi.MARGIN_OLD:
            R0, P0 = _cam_pose(pose[0], ex); R1, P1 = _cam_pose(pose[1], ex)
            fm.removeBackShiftDepth(R0, P0, R1, P1)
            pose[:-1], sb[:-1], stamp[:-1] = pose[1:].copy(), sb[1:].copy(), stamp[1:].copy()
            samples[:] = samples[1:] + [None]; lidar[:] = lidar[1:] + [None]
            reset_interval(K - 1, nxt["first"])
        else:
            fm.removeFront(K - 1)
            a, b = samples[K - 2], samples[K - 1]
            samples[K - 2] = [np.concatenate([a[0], b[0]]), np.vstack([a[1], b[1]]), np.vstack([a[2], b[2]]), a[3], a[4], a[5], a[6]]
            pose[K - 2], sb[K - 2], stamp[K - 2] = pose[K - 1], sb[K - 1], stamp[K - 1]
            lidar[K - 2] = lidar[K - 1]
            reset_interval(K - 1, nxt["first"])
        fm.removeFailures()
        take(K - 1, nxt, False)
    text = "".join(lines)
    if log_path:
        with open(log_path, "w") as f:
            f.write(text)
    if shadow is not None:
        return text, recs, "".join(slines), srecs
    return text, recs

#!/usr/bin/env python3
"""Generates the golden fixtures tests/golden/*.npz.

The reference ships no golden vectors and cannot be compiled or imported here (C++ needing
Eigen/Ceres/ROS -- SURVEY.md 8c), so the fixtures are produced by the CPU restatement in oracle/
(itself pinned by the finite-difference / identity tests) on small deterministic windows.  Each file
holds the INPUT window (every table of vil_problem / vil_state) and the EXPECTED outputs:
per-class residuals and Jacobians, the Schur-reduced normal equations, the solve summary and the
solved + gauge-fixed state, the marginalisation result.   Run from the repo root:
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g  # noqa: E402

g.load_package()
from mvil_fusion_amd import abi, synth  # noqa: E402
import oracle_lib  # noqa: E402

CASES = {
    "c1_mini": (1, dict(L=24)),                                                  # IMU + visual only, no prior (config 1 shape)
    "c2_mini": (2, dict(L=40, n_plane=160, n_edge=60)),                          # VIL window with prior, ICP, LPS, LiDAR points
    "c4_mini": (4, dict(L=36)),                                                  # K = 20 window, prior n = 130
}


def main():
    orc = oracle_lib.open_oracle()
    pf = lambda pre: orc.marginalize(pre).to_prior()
    for name, (cid, kw) in CASES.items():
        w = synth.make_config(cid, prior_fn=pf, **kw)
        out = {"in_" + k: v for k, v in w.to_dict().items()}
        for cls in range(7):
            r, J = orc.eval_factors(w, cls)
            out["r%d" % cls], out["J%d" % cls] = r, J
        cost, S, gv = orc.linearize(w)
        out["lin_cost"], out["lin_S"], out["lin_g"] = np.array([cost]), S, gv
        w2 = abi.Window.from_dict(w.to_dict())
        p0 = w2.pose[0].copy()
        sm = orc.solve(w2)
        orc.gauge_fix(p0, w2)
        out["sol_summary"] = np.array([sm.iterations, sm.successful_steps, sm.termination, sm.initial_cost, sm.final_cost])
        out["sol_trace"] = np.array(sm.cost_trace[:sm.iterations])
        for k, v in w2.state_copy().items():
            out["sol_" + k] = v
        mo = orc.marginalize(w2, abi.MARGIN_OLD)
        out["marg_meta"] = np.array([mo.c.n, mo.c.nblk, mo.c.m]); out["marg_A"], out["marg_b"] = mo.A_matrix(), mo.b_vector()
        out["marg_kind"], out["marg_index"], out["marg_col"] = mo.blk_kind[:mo.c.nblk].copy(), mo.blk_index[:mo.c.nblk].copy(), mo.blk_col[:mo.c.nblk].copy()
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(name, "K=%d L=%d Fv=%d" % (w.K, w.L, len(w.vis_i)), "%.0f KB" % (os.path.getsize(path) / 1024))
    make_vgicp(orc)
    make_mapreg(orc)


def make_mapreg(orc):
    """tests/golden/mapreg/mapreg_mini.npz: a small local map + scan (inputs), the oracle's correspondences at a fixed pose and
    the registered pose of the two-round alignment (expected outputs).  SURVEY 8(f) row 2."""
    from mvil_fusion_amd import mapreg
    from mvil_fusion_amd.vgicp import _rot
    cm, sm = mapreg.make_map(seed=31, n_surf=3000, n_corner=700)
    R, t = _rot(0.008, -0.012, 0.35), np.array([0.8, -1.2, 0.2])
    sc, ss = mapreg.make_scan(cm, sm, R, t, seed=32, n_surf=400, n_corner=120)
    q0 = mapreg.quat_from_R(R @ _rot(0.004, -0.003, 0.008)); t0 = t + np.array([0.05, -0.04, 0.03])
    reg = mapreg.MapReg(orc.lib, "orc_vmap_")
    reg.set_map(cm, sm)
    edge, plane = reg.associate(sc, ss, q0, t0)
    q, tt, s = reg.align(None, sc, ss, q0, t0)
    out = dict(map_corner=cm, map_surf=sm, scan_corner=sc, scan_surf=ss, q0=q0, t0=t0, q_true=mapreg.quat_from_R(R), t_true=t, edge=edge, plane=plane,
               align_q=q, align_t=tt, align_meta=np.array([s.rounds, s.n_edge, s.n_plane, s.iterations, s.initial_cost, s.final_cost]))
    os.makedirs(os.path.join(HERE, "mapreg"), exist_ok=True)
    path = os.path.join(HERE, "mapreg", "mapreg_mini.npz")
    np.savez_compressed(path, **out)
    print("mapreg_mini", "%d+%d map / %d+%d scan points, %d edge + %d plane factors" % (len(cm), len(sm), len(sc), len(ss), len(edge), len(plane)), "%.0f KB" % (os.path.getsize(path) / 1024))


def make_fast_gicp_kat(_orc=None):
    """tests/golden/vgicp/fast_gicp_kat.npz: the KNOWN-ANSWER data of the reference's own registration test -- the vendored
    fast_gicp ships `src/test/gicp_test.cpp` with two real LiDAR scans and their relative pose
    (vils_estimator/src/lidar_functions/fast_gicp-master.zip: data/251370668.pcd, data/251371071.pcd, data/relative.txt).
    The test down-samples both scans with a 0.2 m voxel grid (centroid per voxel, pcl::VoxelGrid) and requires every
    registration to land within 0.05 m / 1 degree of relative.txt and to report convergence (gicp_test.cpp:43-60, :133-150).
    The fixture holds the down-sampled scans (float32) and the pose; only this script reads /root/reference."""
    import io, zipfile
    z = zipfile.ZipFile("/root/reference/vils_estimator/src/lidar_functions/fast_gicp-master.zip")

    def load_pcd(name):
        raw = z.read("fast_gicp-master/data/" + name)
        head, body = raw.split(b"DATA binary\n", 1)
        fields = dict(l.split(" ", 1) for l in head.decode().splitlines() if " " in l)
        assert fields["FIELDS"] == "x y z intensity" and fields["SIZE"] == "4 4 4 4" and fields["TYPE"] == "F F F F"
        n = int(fields["POINTS"])
        return np.frombuffer(body[:16 * n], np.float32).reshape(n, 4)[:, :3].copy()

    def voxel_grid(xyz, leaf):
        # pcl::VoxelGrid: voxel = floor(p / leaf), centroid of the points of a voxel, output ordered by the linear voxel index
        xyz = xyz[np.all(np.isfinite(xyz), axis=1)]
        ijk = np.floor(xyz.astype(np.float64) / leaf).astype(np.int64)
        ijk -= ijk.min(axis=0)
        dim = ijk.max(axis=0) + 1
        lin = ijk[:, 0] + ijk[:, 1] * dim[0] + ijk[:, 2] * dim[0] * dim[1]
        order = np.argsort(lin, kind="stable")
        lin_s, pts = lin[order], xyz[order].astype(np.float64)
        first = np.concatenate([[True], lin_s[1:] != lin_s[:-1]])
        grp = np.cumsum(first) - 1
        cnt = np.bincount(grp)
        cen = np.stack([np.bincount(grp, weights=pts[:, c]) / cnt for c in range(3)], axis=1)
        return cen.astype(np.float32)

    rel = np.array([float(v) for v in z.read("fast_gicp-master/data/relative.txt").decode().split()]).reshape(4, 4)
    tgt, src = voxel_grid(load_pcd("251370668.pcd"), 0.2), voxel_grid(load_pcd("251371071.pcd"), 0.2)
    os.makedirs(os.path.join(HERE, "vgicp"), exist_ok=True)
    path = os.path.join(HERE, "vgicp", "fast_gicp_kat.npz")
    np.savez_compressed(path, target=tgt, source=src, relative_pose=rel, t_tol=np.array([0.05]), r_tol_deg=np.array([1.0]))
    print("fast_gicp_kat", "%d target / %d source points after the 0.2 m voxel grid" % (len(tgt), len(src)), "%.0f KB" % (os.path.getsize(path) / 1024))


def make_vgicp(orc):
    """tests/golden/vgicp/vgicp_mini.npz: a small scan pair (inputs) + the oracle's linearisation at a fixed transform for
    the three neighbour modes and the aligned transform of both optimisers (expected outputs).  SURVEY 8(f) row 1."""
    from mvil_fusion_amd import vgicp
    tx, tc, sx, sc, T_true = vgicp.make_pair(seed=11, rings=6, az=240)
    out = dict(tgt_xyz=tx, tgt_cov=tc, src_xyz=sx, src_cov=sc, T_true=T_true, resolution=np.array([0.5]))
    reg = vgicp.Vgicp(orc.lib, "orc_vgicp_")
    reg.set_target(tx, tc, 0.5); reg.set_source(sx, sc)
    T = np.eye(4); T[:3, :3] = vgicp._rot(0.003, -0.002, 0.008); T[:3, 3] = [0.04, -0.03, 0.01]
    out["T_lin"] = T
    for mode in (vgicp.DIRECT1, vgicp.DIRECT7, vgicp.DIRECT27):
        e, H, b, n = reg.linearize(T, mode)
        out["lin%d_err" % mode], out["lin%d_H" % mode], out["lin%d_b" % mode], out["lin%d_n" % mode] = np.array([e]), H, b, np.array([n])
    for name, opt in (("lm", vgicp.LM), ("gn", vgicp.GN)):
        Ta, s = reg.align(np.eye(4), reg.default_options(optimizer=opt))
        out["align_%s_T" % name] = Ta
        out["align_%s_meta" % name] = np.array([s.iterations, s.converged, s.n_correspondences, s.final_error])
    os.makedirs(os.path.join(HERE, "vgicp"), exist_ok=True)
    path = os.path.join(HERE, "vgicp", "vgicp_mini.npz")
    np.savez_compressed(path, **out)
    print("vgicp_mini", "%d target / %d source points" % (len(tx), len(sx)), "%.0f KB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__":
    if len(sys.argv) > 1:                      # e.g. `make_golden.py mapreg`: only that fixture
        {"vgicp": make_vgicp, "mapreg": make_mapreg, "fast_gicp_kat": make_fast_gicp_kat}[sys.argv[1]](oracle_lib.open_oracle())
    else:
        main()

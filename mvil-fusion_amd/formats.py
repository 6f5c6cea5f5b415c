"""Python mirror of include/vilformat.hpp: the /feature_tracker_/feature PointCloud channel layout
(feature_tracker_node.cpp:127-177, estimator_node.cpp:485-503) and the Frontend.txt trajectory log (visualization.cpp:199-212)."""
import numpy as np


def decode_feature_cloud(points_xyz, channels, num_of_cam=1, first_camera_only=True):
    """points_xyz: n x 3 float32, channels: 6 float32 arrays (id * NUM_OF_CAM + cam, u, v, vx, vy, depth).
    Returns (ids, camera_ids, obs8) sorted by feature id like the reference's std::map."""
    pts = np.asarray(points_xyz, np.float32).reshape(-1, 3)
    ch = [np.asarray(c, np.float32) for c in channels]
    v = (ch[0] + np.float32(0.5)).astype(np.int64)
    fid, cam = v // num_of_cam, v % num_of_cam
    if np.any(pts[:, 2] != 1.0):
        raise ValueError("feature point with z != 1")
    order = np.argsort(fid, kind="stable")
    if first_camera_only and len(order):
        keep = np.concatenate([[True], fid[order][1:] != fid[order][:-1]])
        order = order[keep]
    obs8 = np.column_stack([pts[order].astype(np.float64)] + [c[order].astype(np.float64) for c in ch[1:]])
    return fid[order].astype(np.int32), cam[order].astype(np.int32), obs8


def encode_feature_cloud(ids, camera_ids, obs8, num_of_cam=1):
    obs8 = np.asarray(obs8, np.float64).reshape(-1, 8)
    pts = np.column_stack([obs8[:, 0], obs8[:, 1], np.ones(len(obs8))]).astype(np.float32)
    ch = [(np.asarray(ids) * num_of_cam + np.asarray(camera_ids)).astype(np.float32)] + [obs8[:, c].astype(np.float32) for c in (3, 4, 5, 6, 7)]
    return pts, ch


def format_trajectory_line(stamp, P, q_xyzw):
    return "%.9f %.5f %.5f %.5f %.5f %.5f %.5f %.5f\n" % (stamp, P[0], P[1], P[2], q_xyzw[0], q_xyzw[1], q_xyzw[2], q_xyzw[3])


def parse_trajectory(text):
    """-> n x 8 array [stamp px py pz qx qy qz qw]"""
    rows = [[float(x) for x in l.split()] for l in text.splitlines() if l.strip()]
    return np.array(rows).reshape(-1, 8)

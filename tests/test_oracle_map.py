"""Pins of the CPU restatement of lidar_mapping's scan-to-map association + solve (oracle/oracle_map.cpp): numpy
re-derivation of single correspondences (kd-tree semantics, PCA line test, intensity re-ranking, plane fit), geometric sanity
of every emitted factor, and recovery of a known pose offset by the two-round alignment."""
import numpy as np
import pytest

from mvil_fusion_amd import abi, mapreg
from mvil_fusion_amd.vgicp import _rot


@pytest.fixture(scope="module")
def scene():
    cm, sm = mapreg.make_map(seed=2, n_surf=6000, n_corner=1200)
    R, t = _rot(0.01, -0.02, 0.4), np.array([1.0, -2.0, 0.3])
    sc, ss = mapreg.make_scan(cm, sm, R, t, seed=3, n_surf=800, n_corner=200)
    return cm, sm, sc, ss, R, t


@pytest.fixture()
def reg(oracle, scene):
    r = mapreg.MapReg(oracle.lib, "orc_vmap_")
    r.set_map(scene[0], scene[1])
    yield r
    r.close()


def test_association_matches_numpy(reg, scene):
    cm, sm, sc, ss, R, t = scene
    q = mapreg.quat_from_R(R)
    edge, plane = reg.associate(sc, ss, q, t)
    assert len(edge) > 50 and len(plane) > 400
    # every emitted edge: |a - b| = 0.2, the scan point mapped with the true pose lies within a few cm of the line
    assert np.allclose(np.linalg.norm(edge[:, 3:6] - edge[:, 6:9], axis=1), 0.2, atol=1e-12)
    pw = edge[:, :3] @ R.T + t
    d = np.linalg.norm(np.cross(pw - edge[:, 3:6], pw - edge[:, 6:9]), axis=1) / 0.2
    assert np.median(d) < 0.1
    # every emitted plane: unit normal, point-to-plane distance of the mapped scan point small
    assert np.allclose(np.linalg.norm(plane[:, 3:6], axis=1), 1.0, atol=1e-12)
    dist = np.abs(np.einsum("ij,ij->i", plane[:, 3:6], plane[:, :3] @ R.T + t) + plane[:, 6])
    assert np.median(dist) < 0.06
    # numpy re-derivation of the first few surf correspondences
    Rq = R
    for row in plane[:5]:
        i = int(np.where(np.all(np.isclose(ss[:, :3], row[:3].astype(np.float32)), axis=1))[0][0])
        s = (Rq @ ss[i, :3].astype(np.float64) + t).astype(np.float32)
        d2 = ((s - sm[:, :3]) ** 2).sum(axis=1, dtype=np.float32)
        nn = np.argsort(d2, kind="stable")[:10]
        rk = sorted(nn, key=lambda j: (abs(np.float32(sm[j, 3]) - np.float32(ss[i, 3])), j))[:5]
        A = sm[rk, :3].astype(np.float64)
        n = np.linalg.lstsq(A, -np.ones(5), rcond=None)[0]
        assert np.allclose(n / np.linalg.norm(n), row[3:6], atol=1e-9) and abs(1 / np.linalg.norm(n) - row[6]) < 1e-9


def test_align_recovers_pose(reg, scene):
    cm, sm, sc, ss, R, t = scene
    R0 = R @ _rot(0.004, -0.003, 0.01); t0 = t + np.array([0.06, -0.05, 0.03])
    q, tt, s = reg.align(None, sc, ss, mapreg.quat_from_R(R0), t0)
    assert s.rounds == 2 and s.n_edge > 50 and s.n_plane > 400 and 1 <= s.iterations <= 4
    assert np.linalg.norm(tt - t) < 0.02 and np.linalg.norm(tt - t) < np.linalg.norm(t0 - t)
    assert s.final_cost < s.initial_cost and abs(np.linalg.norm(q) - 1) < 1e-12


def test_too_small_map_is_skipped(oracle, scene):
    r = mapreg.MapReg(oracle.lib, "orc_vmap_")
    r.set_map(scene[0][:8], scene[1][:40])                      # localMapping.cpp:586: needs > 10 corner and > 50 surf map points
    q0, t0 = mapreg.quat_from_R(scene[4]), scene[5]
    q, t, s = r.align(None, scene[2], scene[3], q0, t0)
    r.close()
    assert s.rounds == 0 and np.array_equal(q, q0) and np.array_equal(t, t0)


def test_kdtree_search_equals_exhaustive(oracle, scene):
    """The kd-tree used for the CPU baseline returns exactly the exhaustive result (distances AND tie order)."""
    import ctypes as C
    cm, sm, sc, ss, R, t = scene
    q = mapreg.quat_from_R(R)
    a = mapreg.MapReg(oracle.lib, "orc_vmap_"); b = mapreg.MapReg(oracle.lib, "orc_vmap_")
    a.set_map(cm, sm); b.set_map(cm, sm)
    oracle.lib.orc_vmap_set_search(b.ctx, C.c_int32(1))
    ea, pa = a.associate(sc, ss, q, t); eb, pb = b.associate(sc, ss, q, t)
    assert np.array_equal(ea, eb) and np.array_equal(pa, pb)
    # duplicated map points (exact distance ties): still identical
    cm2, sm2 = np.concatenate([cm, cm[:300]]), np.concatenate([sm, sm[:1500]])
    a.set_map(cm2, sm2); b.set_map(cm2, sm2)
    ea, pa = a.associate(sc, ss, q, t); eb, pb = b.associate(sc, ss, q, t)
    assert np.array_equal(ea, eb) and np.array_equal(pa, pb)
    a.close(); b.close()


def _check_golden(reg, solver_ctx):
    import os
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mapreg", "mapreg_mini.npz"))
    reg.set_map(d["map_corner"], d["map_surf"])
    edge, plane = reg.associate(d["scan_corner"], d["scan_surf"], d["q0"], d["t0"])
    assert edge.shape == d["edge"].shape and plane.shape == d["plane"].shape
    assert np.array_equal(edge[:, :3], d["edge"][:, :3]) and np.array_equal(plane[:, :3], d["plane"][:, :3])
    same = np.abs(edge[:, 3:] - d["edge"][:, 3:]).max(axis=1) < 1e-9
    swapped = np.abs(edge[:, 3:6] - d["edge"][:, 6:9]).max(axis=1) + np.abs(edge[:, 6:9] - d["edge"][:, 3:6]).max(axis=1) < 1e-9
    assert np.all(same | swapped) and np.abs(plane[:, 3:] - d["plane"][:, 3:]).max() < 1e-9
    q, t, s = reg.align(solver_ctx, d["scan_corner"], d["scan_surf"], d["q0"], d["t0"])
    m = d["align_meta"]
    assert (s.rounds, s.n_edge, s.n_plane, s.iterations) == tuple(int(v) for v in m[:4])
    assert abs(s.final_cost - m[5]) <= 1e-8 * m[5] and np.abs(q - d["align_q"]).max() < 1e-9 and np.abs(t - d["align_t"]).max() < 1e-8


def test_oracle_reproduces_golden_fixture(oracle):
    r = mapreg.MapReg(oracle.lib, "orc_vmap_")
    _check_golden(r, None)
    r.close()

"""Window residency across frames (include/vilsolve.h: vil_lidar_push / drop, vil_set_gauge_fix, vil_marginalize_resident):
the per-image chain solve -> gauge fix -> marginalise -> slide driven through the resident entry points reproduces the chain
driven through vil_solve / vil_gauge_fix / vil_marginalize with every table handed over on every image."""
import numpy as np
import pytest

from mvil_fusion_amd import abi, lib, replay

pytestmark = pytest.mark.gpu


def run_classic(be, rp, n):
    out = []
    for _ in range(n):
        w = rp.window(); flag = rp.margin_flag()
        p0 = w.pose[0].copy()
        sm = be.solve(w, rp.opts); be.gauge_fix(p0, w)
        pg = be.marginalize(w, flag, w._icp_marg, w._lps_marg, rp.opts)
        out.append((sm.iterations, w.pose.copy(), w.speedbias.copy(), w.inv_depth.copy(), pg.A_matrix() if pg.c.n > 0 else None, pg.c.n))
        if not rp.absorb(w, pg, flag):
            break
    return out


def run_resident(be, rp, n):
    K = rp.K
    be.set_gauge_fix(True); be.lidar_reset()
    for k in range(K):
        be.lidar_push(rp.lidar[k][0], rp.lidar[k][1])
    out = []
    for _ in range(n):
        w = rp.window(with_lidar=False); flag = rp.margin_flag()
        sm = be.solve(w, rp.opts)                                # comes back gauge-fixed
        pg = be.marginalize_resident(w, flag, w._icp_marg, w._lps_marg, rp.opts)
        out.append((sm.iterations, w.pose.copy(), w.speedbias.copy(), w.inv_depth.copy(), pg.A_matrix() if pg.c.n > 0 else None, pg.c.n))
        be.lidar_drop(0 if flag == abi.MARGIN_OLD else K - 2)   # slideWindow on the device: an index remap
        if not rp.absorb(w, pg, flag):
            break
        be.lidar_push(rp.lidar[K - 1][0], rp.lidar[K - 1][1])   # the only LiDAR bytes of this image
    be.set_gauge_fix(False); be.lidar_reset()
    return out


def test_resident_chain_equals_classic_chain():
    kw = dict(K=10, n_frames=34, L=200, n_plane=4000, n_edge=1200, seed=20240611, max_iterations=8)
    be = lib.open_vilsolve()
    a = run_classic(be, replay.Replay(**kw), 24)
    b = run_resident(be, replay.Replay(**kw), 24)
    be.close()
    assert len(a) == len(b) == 24
    flags = set()
    for f, (ra, rb) in enumerate(zip(a, b)):
        assert ra[0] == rb[0], (f, ra[0], rb[0])                                         # same iteration count on every image
        assert np.abs(ra[1] - rb[1]).max() < 1e-8 and np.abs(ra[2] - rb[2]).max() < 1e-7, (f, np.abs(ra[1] - rb[1]).max())
        assert np.abs(ra[3] - rb[3]).max() < 1e-7
        assert ra[5] == rb[5]
        flags.add(ra[5] > 0)
        if ra[4] is not None:
            sc = np.sqrt(np.maximum(np.abs(np.diag(ra[4])), 1e-300))
            assert np.abs((ra[4] - rb[4]) / np.outer(sc, sc)).max() < 2e-5, f           # (the tolerance of test_gpu_replay: the marginal's fp64 noise floor)
    assert True in flags


def test_resident_entry_points_reject_misuse():
    be = lib.open_vilsolve()
    rp = replay.Replay(K=10, n_frames=14, L=60, n_plane=1000, n_edge=300, seed=3, max_iterations=4)
    w = rp.window(with_lidar=False)
    with pytest.raises(lib.VilError):
        be.marginalize_resident(w)                           # nothing resident yet
    be.lidar_reset()
    for k in range(11):                                      # more slabs than window frames
        be.lidar_push(rp.lidar[k % 10][0], rp.lidar[k % 10][1])
    with pytest.raises(lib.VilError):
        be.solve(w, rp.opts)
    with pytest.raises(lib.VilError):
        be.lidar_drop(11)
    be.lidar_drop(0)
    be.solve(w, rp.opts)                                     # ten slabs: fine
    be.close()

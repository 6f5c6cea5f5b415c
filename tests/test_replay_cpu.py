"""Synthetic replay (config 5) bookkeeping, driven by the CPU oracle: windows stay well-formed across MARGIN_OLD and
MARGIN_SECOND_NEW slides, the prior chain has the right shape, the estimate stays near the truth."""
import numpy as np

from mvil_fusion_amd import abi, replay


def test_replay_chain_oracle(oracle):
    K = 7
    rp = replay.Replay(K=K, n_frames=40, L=60, n_plane=700, n_edge=210, seed=11, second_new_every=4)
    seen = set()

    def check(rp_, w, po, rec):
        assert w.K == K and len(w.vis_i) == len(w.vis_j) == len(w.vis_l) == len(w.vis_const)
        assert w.vis_l.max() < w.L and np.all(np.bincount(w.vis_l, minlength=w.L) >= 1)          # every landmark has >= 1 factor (>= 2 observations)
        assert np.all(w.vis_i < w.vis_j) and w.vis_j.max() <= K - 1 and w.vis_i.max() < K - 3       # feature_manager.cpp:36
        assert np.all(np.diff(w.plane_pose) >= 0) and w.plane_pose.max() == K - 1
        assert len(w.imu_i) == K - 1 and np.all(w.imu_const[:, 16] > 0.09)
        assert len(w.icp_ids) <= 5 and len(w.lps_ids) <= 7
        assert np.isfinite(rec["final_cost"]) and rec["final_cost"] <= rec["initial_cost"]
        seen.add(rec["flag"])
        if rec["flag"] == abi.MARGIN_OLD:
            kinds = list(po.blk_kind[:po.c.nblk]); idx = list(po.blk_index[:po.c.nblk])
            poses = sorted(i for k, i in zip(kinds, idx) if k == abi.BLK_POSE)
            # kept: the poses the factors of frame 0 (and the old prior) touch, shifted down by one; speed-bias 1 -> 0; ex; td
            assert set(poses) <= set(range(K - 1)) and 0 in poses and po.c.n == 6 * len(poses) + 9 + 6 + 1
        else:
            assert po.c.n in (-1, 6 * (K - 2) + 9 + 6 + 1)                                        # pose K-2 dropped from the prior (or prior untouched)

    import os, tempfile
    from mvil_fusion_amd import formats
    log = os.path.join(tempfile.mkdtemp(), "Frontend.txt")
    recs = replay.run(oracle, rp, n_steps=14, on_frame=check, log_path=log)
    assert len(recs) == 14 and seen == {abi.MARGIN_OLD, abi.MARGIN_SECOND_NEW}
    traj = formats.parse_trajectory(open(log).read())                                             # the reference's trajectory log format
    assert traj.shape == (14, 8) and np.all(np.diff(traj[:, 0]) > 0) and np.allclose(np.linalg.norm(traj[:, 4:], axis=1), 1.0, atol=2e-5)
    assert recs[0]["prior_n"] == 0 and all(r["prior_n"] > 0 for r in recs[1:])
    # merged pre-integration after a MARGIN_SECOND_NEW slide spans two keyframe intervals
    assert max(r["pos_err_newest"] for r in recs) < 0.5
    w = rp.window()
    assert any(abs(dt - 0.2) < 1e-9 for dt in w.imu_const[:, 16]) or all(abs(dt - 0.1) < 1e-9 for dt in w.imu_const[:, 16])


def test_replay_deterministic():
    a = replay.Replay(K=6, n_frames=12, L=30, n_plane=60, n_edge=30, seed=3).window()
    b = replay.Replay(K=6, n_frames=12, L=30, n_plane=60, n_edge=30, seed=3).window()
    for k in ("pose", "vis_const", "plane_const", "edge_const", "imu_const", "inv_depth"):
        assert np.array_equal(getattr(a, k), getattr(b, k))

"""The multi-GPU iteration -- sweep + gather into set 0, ONE collective that sums the whole linear-system set into set 1, the
complete step kernel on set 1 -- exercised on ONE GPU: the plumbing forced on a single rank (vil_debug_set_split) with and
without a 1-rank RCCL communicator, then complete factor-sharded solves of 2, 3 and 8 ranks through the in-process communicator."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import sys, os, ctypes as C
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np
import __graft_entry__ as g; g.load_package()
from mvil_fusion_amd import abi, lib, synth
import oracle_lib
orc = oracle_lib.open_oracle()
be = lib.open_vilsolve()
assert be.lib.vil_debug_set_split(be.ctx, 1) == 0
if os.environ.get("USE_COMM") == "1":
    uid = (C.c_char * 128)()
    assert be.lib.vil_comm_unique_id(uid) == 0
    assert be.lib.vil_comm_init(be.ctx, uid, 0, 1) == 0
pf = lambda pre: orc.marginalize(pre).to_prior()
for cid, kw in ((1, {}), (2, dict(L=150, n_plane=3000, n_edge=800))):
    wg = synth.make_config(cid, prior_fn=pf, **kw); wo = synth.make_config(cid, prior_fn=pf, **kw)
    sg, so = be.solve(wg), orc.solve(wo)
    assert sg.iterations == so.iterations and sg.termination == so.termination, (sg.iterations, so.iterations, sg.termination, so.termination)
    assert abs(sg.final_cost - so.final_cost) <= 1e-7 * so.final_cost
    if wg.prior.n:
        assert np.abs(wg.pose - wo.pose).max() < 1e-6
    c1, S1, g1 = be.linearize(wo); c2, S2, g2 = orc.linearize(wo)
    assert abs(c1 - c2) <= 1e-12 * c2 and np.abs(S1 - S2).max() <= 1e-10 * np.abs(S2).max()
print("SPLIT_OK")
'''


@pytest.mark.parametrize("use_comm", ["0", "1"])
def test_forced_split_path_matches_oracle(use_comm):
    env = dict(os.environ, USE_COMM=use_comm)
    out = subprocess.run([sys.executable, "-c", SCRIPT % (ROOT, ROOT)], env=env, capture_output=True, text=True, timeout=600)
    assert "SPLIT_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


SHARD_SCRIPT = r'''
import sys, os, ctypes as C, threading
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np
import __graft_entry__ as g; g.load_package()
from mvil_fusion_amd import abi, lib, synth
import oracle_lib
orc = oracle_lib.open_oracle()
pf = lambda pre: orc.marginalize(pre).to_prior()
full = os.environ.get("SHARD_FULL") == "1"
def shard_bounds(w, world):
    """(landmark range, factor range) of every rank: vil_shard_ranges through the C-ABI"""
    be0 = lib.open_vilsolve(); out = []
    p = w.c_problem()
    for r in range(world):
        v = [C.c_int32(0) for _ in range(6)]
        assert be0.lib.vil_shard_ranges(C.byref(p), r, world, *[C.byref(x) for x in v]) == 0
        lb0, lb1 = v[0].value, v[1].value
        f0 = int(np.searchsorted(w.vis_l, lb0, "left")); f1 = int(np.searchsorted(w.vis_l, lb1, "left"))
        out.append((lb0, lb1, f0, f1))
    be0.close()
    return out
for world in ((8,) if full else (2, 3)):
    bes = [lib.open_vilsolve() for _ in range(world)]
    arr = (C.c_void_p * world)(*[b.ctx for b in bes])
    assert bes[0].lib.vil_comm_init_local(arr, world) == 0
    slim = os.environ.get("SHARD_SLIM") == "1"          # the RCCL path's pack -> all-reduce + all-gather -> unpack, the two RCCL calls emulated in process
    if slim:
        for b in bes: assert b.lib.vil_debug_set_slim_emul(b.ctx, 1) == 0
    for cid, kw in (((3, {}), (2, {})) if full else ((2, dict(L=150, n_plane=3000, n_edge=800)), (1, {}))):
        wo = synth.make_config(cid, prior_fn=pf, **kw)
        if full and cid == 3: assert (wo.K, wo.L, len(wo.plane_pose) + len(wo.edge_pose)) == (10, 4000, 120000)     # BASELINE.json configs[2]
        w1 = synth.make_config(cid, prior_fn=pf, **kw)
        be1 = lib.open_vilsolve(); s1 = be1.solve(w1); be1.close()                      # the un-sharded solve of the same window
        ws = [synth.make_config(cid, prior_fn=pf, **kw) for _ in range(world)]
        res = [None] * world
        def run(r):
            try:
                lin = bes[r].linearize(ws[r])                                           # at the initial state, before the solve moves it
                res[r] = ("ok", bes[r].solve(ws[r]), lin)                              # every rank: its shard, the same reductions
            except Exception as e:
                res[r] = ("err", repr(e))
        th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        [t.start() for t in th]; [t.join(300) for t in th]
        assert all(x is not None and x[0] == "ok" for x in res), res
        co, So, go = orc.linearize(wo)
        so = orc.solve(wo)
        for r in range(world):
            sg, (cg, Sg, gg) = res[r][1], res[r][2]
            assert sg.iterations == so.iterations and sg.termination == so.termination, (world, cid, r, sg.iterations, so.iterations)
            assert abs(sg.final_cost - so.final_cost) <= 1e-7 * so.final_cost
            if wo.prior.n:
                assert np.abs(ws[r].pose - wo.pose).max() < 1e-6 and np.abs(ws[r].inv_depth - wo.inv_depth).max() < 1e-6
            assert np.array_equal(ws[r].pose, ws[0].pose) and np.array_equal(ws[r].inv_depth, ws[0].inv_depth)     # ranks agree bit for bit
            assert np.array_equal(ws[r].speedbias, ws[0].speedbias) and np.array_equal(ws[r].ex_pose, ws[0].ex_pose)
            # sharded vs un-sharded on the same device: same iterations, states equal to summation-order rounding
            assert sg.iterations == s1.iterations and abs(sg.final_cost - s1.final_cost) <= 1e-10 * s1.final_cost
            if wo.prior.n:      # (a prior-less window has a 4-dof gauge null space)
                assert np.abs(ws[r].pose - w1.pose).max() < 1e-9 and np.abs(ws[r].inv_depth - w1.inv_depth).max() < 1e-8
            assert abs(cg - co) <= 1e-11 * co and np.abs(Sg - So).max() <= 1e-9 * np.abs(So).max()
            # the per-iteration message of this rank to each peer: lower triangle of S' + vectors + its OWN slice of the landmark arrays (SURVEY 8e asks
            # for ~100 kB: that is the camera part; the landmark arrays are dealt, not replicated)
            mb, fb = C.c_int64(0), C.c_int64(0)
            assert bes[r].lib.vil_comm_message_bytes(bes[r].ctx, C.byref(mb), C.byref(fb)) == 0
            D = 15 * wo.K + 7; cam = 8 * (D * (D + 1) // 2 + 3 * D + 4); lmk = 8 * (17 * wo.L + 6 * len(wo.vis_i))
            assert cam <= mb.value <= cam + 1.6 * lmk / world + 1024 and fb.value >= 8 * D * D + lmk, (world, r, mb.value, fb.value)
            if full and cid == 2: assert mb.value <= 160 * 1024, mb.value            # configs[1] on 8 ranks: ~150 kB per peer (484 kB as one all-reduced set)
            if slim:      # what RCCL moves per rank: the packed camera part (all-reduce) + the largest owner slice (all-gather)
                own = [17 * (lb1 - lb0) + 6 * (fb1 - fb0) for (lb0, lb1, fb0, fb1) in shard_bounds(wo, world)]
                assert mb.value == 8 * (((D * (D + 1) // 2 + 3 * D + 4 + 1) // 2) * 2 + ((max(own) + 2) // 2) * 2), (mb.value, D, max(own))
        # sharded marginalisation (SURVEY 8e last row): the collected factors dealt to the ranks, A / b all-reduced once, the small dense part on
        # every rank -- equal to the un-sharded marginal of the same (solved) window, bit-identical across ranks
        if wo.prior.n and not full:
            be1 = lib.open_vilsolve(); p1 = be1.marginalize(w1, abi.MARGIN_OLD, 0, 0); be1.close()
            mres = [None] * world
            def mrun(r):
                try: mres[r] = ("ok", bes[r].marginalize(ws[r], abi.MARGIN_OLD, 0, 0))
                except Exception as e: mres[r] = ("err", repr(e))
            th = [threading.Thread(target=mrun, args=(r,)) for r in range(world)]
            [t.start() for t in th]; [t.join(300) for t in th]
            assert all(x is not None and x[0] == "ok" for x in mres), mres
            A1 = p1.A_matrix(); sc = np.sqrt(np.outer(np.abs(np.diag(A1)) + 1e-300, np.abs(np.diag(A1)) + 1e-300))
            for r in range(world):
                pr = mres[r][1]
                assert pr.c.n == p1.c.n and np.array_equal(pr.blk_kind[:pr.c.nblk], p1.blk_kind[:p1.c.nblk])
                assert (np.abs(pr.A_matrix() - A1) / sc).max() < 2e-5, (world, cid, r)
                assert np.array_equal(pr.J0[:pr.c.n ** 2], mres[0][1].J0[:pr.c.n ** 2]) and np.array_equal(pr.r0[:pr.c.n], mres[0][1].r0[:pr.c.n])
    for b in bes: b.close()
print("SHARD_OK")
'''


RESIDENT_SCRIPT = r'''
import sys, os, ctypes as C, threading
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np
import __graft_entry__ as g; g.load_package()
from mvil_fusion_amd import abi, lib, replay
kw = dict(K=8, n_frames=24, L=120, n_plane=2400, n_edge=800, seed=31, max_iterations=6, second_new_every=4)
def chain(bes, n):
    """the resident-slab chain (vil_lidar_push / vil_solve / vil_marginalize_resident) driven on every rank of `bes` in lockstep"""
    world = len(bes)
    rps = [replay.Replay(**kw) for _ in range(world)]
    out = [[] for _ in range(world)]
    def run(r):
        be, rp = bes[r], rps[r]
        be.set_gauge_fix(True); be.lidar_reset()
        for k in range(rp.K): be.lidar_push(rp.lidar[k][0], rp.lidar[k][1])
        for _ in range(n):
            w = rp.window(with_lidar=False); flag = rp.margin_flag()
            sm = be.solve(w, rp.opts)
            pg = be.marginalize_resident(w, flag, w._icp_marg, w._lps_marg, rp.opts)
            out[r].append((sm.iterations, w.pose.copy(), w.inv_depth.copy(), pg.A_matrix() if pg.c.n > 0 else None))
            be.lidar_drop(0 if flag == abi.MARGIN_OLD else rp.K - 2)
            assert rp.absorb(w, pg, flag)
            be.lidar_push(rp.lidar[rp.K - 1][0], rp.lidar[rp.K - 1][1])
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]; [t.join(600) for t in th]
    return out
one = chain([lib.open_vilsolve()], 10)[0]
bes = [lib.open_vilsolve() for _ in range(2)]
arr = (C.c_void_p * 2)(*[b.ctx for b in bes])
assert bes[0].lib.vil_comm_init_local(arr, 2) == 0
two = chain(bes, 10)
assert len(two[0]) == len(two[1]) == len(one) == 10
for f in range(10):
    a, b0, b1 = one[f], two[0][f], two[1][f]
    assert a[0] == b0[0] == b1[0], (f, a[0], b0[0], b1[0])
    assert np.array_equal(b0[1], b1[1]) and np.array_equal(b0[2], b1[2])                     # ranks agree bit for bit
    assert np.abs(a[1] - b0[1]).max() < 1e-8 and np.abs(a[2] - b0[2]).max() < 1e-7, (f, np.abs(a[1] - b0[1]).max())
    assert (a[3] is None) == (b0[3] is None)
    if a[3] is not None:
        sc = np.sqrt(np.maximum(np.abs(np.diag(a[3])), 1e-300))
        assert np.abs((a[3] - b0[3]) / np.outer(sc, sc)).max() < 2e-5 and np.array_equal(b0[3], b1[3])
print("RESIDENT_SHARD_OK")
'''


WINDOW_SCRIPT = r'''
import sys, os, ctypes as C, threading
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np
import __graft_entry__ as g; g.load_package()
from mvil_fusion_amd import abi, lib, replay
kw = dict(K=10, n_frames=32, L=200, n_plane=4000, n_edge=1200, seed=20240611, max_iterations=8, second_new_every=5)
N = 16
def chain(bes):
    """the fully resident window (vil_win_*) driven on every rank of `bes` in lockstep: every rank is handed every frame and every small table"""
    world = len(bes)
    rps = [replay.Replay(**kw) for _ in range(world)]
    out = [[] for _ in range(world)]
    err = [None] * world
    def run(r):
        try:
            be, rp = bes[r], rps[r]
            be.set_gauge_fix(False)                                   # vil_win_solve runs the device gauge fix whatever this is left at
            be.win_open(**rp.win_open_args())
            for k in range(rp.K): be.win_push_frame(rp.win_frame(k))
            for _ in range(N):
                w = rp.win_window(); flag = rp.margin_flag()
                sm = be.win_solve(w, rp.opts)
                info = be.win_marginalize(flag, w._icp_marg, w._lps_marg, rp.opts)
                pg = be.win_prior_download(rp.K)
                msg = C.c_int64(0); full = C.c_int64(0)
                if world > 1: assert be.lib.vil_comm_message_bytes(be.ctx, C.byref(msg), C.byref(full)) == 0
                out[r].append((sm.iterations, sm.termination, w.pose.copy(), w.inv_depth.copy(), pg.A_matrix() if pg.c.n > 0 else None, int(info.n), flag, msg.value, full.value))
                be.win_drop_frame(flag)
                assert rp.absorb(w, None, flag)
                be.win_push_frame(rp.win_frame(rp.K - 1))
        except BaseException as e:
            err[r] = e; raise
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]; [t.join(900) for t in th]
    assert all(e is None for e in err), err
    return out
one = chain([lib.open_vilsolve()])[0]
for world in (2, 3):
    bes = [lib.open_vilsolve() for _ in range(world)]
    arr = (C.c_void_p * world)(*[b.ctx for b in bes])
    assert bes[0].lib.vil_comm_init_local(arr, world) == 0
    many = chain(bes)
    assert all(len(m) == N for m in many) and len(one) == N
    flags = set()
    for f in range(N):
        a = one[f]; flags.add(a[6])
        for r in range(world):
            b = many[r][f]
            assert (a[0], a[1], a[5]) == (b[0], b[1], b[5]), (world, f, r, a[0], b[0], a[5], b[5])
            assert np.array_equal(b[2], many[0][f][2]) and np.array_equal(b[3], many[0][f][3])           # ranks agree bit for bit
            # (a CHAIN of 16 images: the summation-order difference of one image's solve reaches the next through the prior -- the tolerances of test_resident_window_equals_classic_entry_points)
            assert np.abs(a[2] - b[2]).max() < 1e-7 and np.abs(a[3] - b[3]).max() < 1e-6, (world, f, np.abs(a[2] - b[2]).max(), np.abs(a[3] - b[3]).max())
            assert (a[4] is None) == (b[4] is None)
            if a[4] is not None:
                sc = np.sqrt(np.maximum(np.abs(np.diag(a[4])), 1e-300))
                assert np.abs((a[4] - b[4]) / np.outer(sc, sc)).max() < 2e-5 and np.array_equal(b[4], many[0][f][4])
            assert 0 < b[7] < 0.62 * b[8], (b[7], b[8])                   # lower(S') + vectors + the OWNED slice of the landmark arrays
    assert flags == {abi.MARGIN_OLD, abi.MARGIN_SECOND_NEW}
    for b in bes: b.close()
print("WINDOW_SHARD_OK")
'''


def test_resident_window_under_a_communicator():
    """vil_win_* on 2 and 3 ranks (in-process communicator, one device): every rank is handed every frame (whole observation store and IMU slots, ITS
    slice of each LiDAR slab) and the same landmark list, keeps the visual factors of its landmark range, and writes the SAME new prior into its own device
    slot from the all-reduced marginalisation system.  16 images with both marginalisation branches: ranks bit-identical, iteration counts, termination
    and prior sizes those of the single-context resident window, states equal to it at summation-order level."""
    out = subprocess.run([sys.executable, "-c", WINDOW_SCRIPT % (ROOT, ROOT)], capture_output=True, text=True, timeout=1200)
    assert "WINDOW_SHARD_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


@pytest.mark.parametrize("slim", ["0", "1"])
def test_sharded_solve_local_communicator(slim):
    """2 and 3 ranks of the factor-sharded solve on ONE device through the in-process communicator: the shard ranges,
    ranks without IMU / prior factors, the split step, the scalar reduction and the landmark merge all run as on N GPUs.
    slim = 1: the RCCL path's packed message (one all-reduce of [lower(S') | vectors | cost] + one all-gather of the owners' landmark
    slices; include/vilsolve.h) with the two RCCL calls emulated over the same communicator."""
    env = dict(os.environ, SHARD_SLIM=slim)
    out = subprocess.run([sys.executable, "-c", SHARD_SCRIPT % (ROOT, ROOT)], env=env, capture_output=True, text=True, timeout=900)
    assert "SHARD_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_sharded_full_size_8_ranks():
    """BASELINE.json configs[2] (K = 10, L = 4000, 120 k LiDAR points) and configs[1] at FULL size as an 8-rank factor-sharded
    solve -- eight contexts, eight host threads, the in-process communicator on one device: every rank bit-identical, equal to
    the oracle and to the un-sharded solve."""
    env = dict(os.environ, SHARD_FULL="1", SHARD_SLIM="1")      # (the packed message of the RCCL path; the plain in-process exchange runs at 2 / 3 ranks above)
    out = subprocess.run([sys.executable, "-c", SHARD_SCRIPT % (ROOT, ROOT)], env=env, capture_output=True, text=True, timeout=1500)
    assert "SHARD_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_sharded_resident_slabs_and_marginalisation():
    """Two ranks (in-process communicator, one device): every rank keeps its slice of each LiDAR frame slab, solves its shard of the window and
    marginalises the RESIDENT shard -- A / b all-reduced once -- over ten images with both marginalisation branches: ranks bit-identical, equal to
    the single-context chain."""
    out = subprocess.run([sys.executable, "-c", RESIDENT_SCRIPT % (ROOT, ROOT)], capture_output=True, text=True, timeout=900)
    assert "RESIDENT_SHARD_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]

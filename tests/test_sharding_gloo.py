"""N>1 path on CPU: world_size-2 gloo processes shard the factor set with vil_shard_ranges (the
library's own partition), each linearises ITS shard with the CPU oracle, the partial normal equations
are all-reduced exactly as the GPU path all-reduces [H | g | cost] over RCCL, and the sum
must equal the un-sharded linearisation."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def shard_window(w, so, rank, world):
    from mvil_fusion_amd import abi
    p = w.c_problem()
    v = [C.c_int32() for _ in range(6)]
    assert so.vil_shard_ranges(C.byref(p), rank, world, *[C.byref(x) for x in v]) == 0
    lb, le, eb, ee, pb, pe = [x.value for x in v]
    import copy
    ws = copy.copy(w)
    m = (w.vis_l >= lb) & (w.vis_l < le)
    ws.vis_i, ws.vis_j, ws.vis_l, ws.vis_const = w.vis_i[m], w.vis_j[m], w.vis_l[m], w.vis_const[m]
    ws.edge_pose, ws.edge_const = w.edge_pose[eb:ee], w.edge_const[eb:ee]
    ws.plane_pose, ws.plane_const = w.plane_pose[pb:pe], w.plane_const[pb:pe]
    if rank != 0:
        ws.imu_i, ws.imu_j, ws.imu_const = w.imu_i[:0], w.imu_j[:0], w.imu_const[:0]
        ws.icp_ids, ws.icp_const, ws.lps_ids, ws.lps_const = w.icp_ids[:0], w.icp_const[:0], w.lps_ids[:0], w.lps_const[:0]
        ws.prior = abi.Prior()
    return ws, (lb, le)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import __graft_entry__ as g
    g.load_package()
    from mvil_fusion_amd import lib, synth
    import oracle_lib
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = oracle_lib.open_oracle()
    so = lib.load_vilsolve()
    w = synth.make_config(2, L=120, n_plane=1500, n_edge=500)
    ws, (lb, le) = shard_window(w, so, rank, world)
    cost, S, gvec = orc.linearize(ws)
    # the Schur complement is additive over landmark owners because every landmark's factors live on ONE rank
    buf = torch.from_numpy(np.concatenate([S.ravel(), gvec, [cost]]))
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    if rank == 0:
        cf, Sf, gf = orc.linearize(w)
        D = w.D
        S_sum, g_sum, c_sum = buf[:D * D].numpy().reshape(D, D), buf[D * D:D * D + D].numpy(), float(buf[-1])
        ok = (abs(c_sum - cf) <= 1e-12 * cf and np.abs(S_sum - Sf).max() <= 1e-11 * np.abs(Sf).max() and np.abs(g_sum - gf).max() <= 1e-11 * np.abs(gf).max())
        q.put(bool(ok))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_linearisation_sums_to_full(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + world + (os.getpid() % 500)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert q.get(timeout=10) is True

"""vil_profile_workgroups: the entry / exit times every workgroup of one chosen launch of a one-launch iteration leaves behind (tools/probe_workgroups.py)."""
import ctypes as C
import numpy as np
import pytest

from mvil_fusion_amd import abi, lib, synth

pytestmark = pytest.mark.gpu


def test_workgroup_times_of_one_launch():
    be = lib.open_vilsolve()
    w = synth.make_config(2, L=200, n_plane=3000, n_edge=800)
    be.upload(w)
    opts = abi.default_options()
    one = C.c_int32(0); lpi = C.c_int32(0)
    assert be.lib.vil_debug_get_launch_structure(be.ctx, C.byref(lpi), C.byref(one)) == 0
    if not one.value: pytest.skip("this device does not take the one-launch iteration for the window")
    be.lib.vil_profile_workgroups.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_uint64), C.c_int32]
    buf = (C.c_uint64 * (2 * 4096))()
    assert be.lib.vil_profile_workgroups(be.ctx, 2, buf, 4096) != 0          # nothing to read before profiling was ever on
    be.lib.vil_profile_enable(be.ctx, 1)
    assert be.lib.vil_profile_workgroups(be.ctx, 2, None, 0) == 0            # arm: launch 2 of the next solves
    s = be.solve_resident(opts)
    assert s.iterations >= 3
    assert be.lib.vil_profile_workgroups(be.ctx, 2, buf, 4096) == 0
    a = np.array(buf[:], dtype=np.uint64).reshape(-1, 2)
    idx = np.nonzero(a[:, 0])[0]
    n_imu = w.K - 1
    assert len(idx) > n_imu + 3 + 8 and idx[0] == 0 and np.array_equal(idx, np.arange(len(idx)))      # every workgroup of the grid, in block order
    tin, tout = a[idx, 0].astype(np.int64), a[idx, 1].astype(np.int64)
    assert (tout >= tin).all()
    stay = (tout - tin) * 0.01                                                                         # us (100 MHz)
    assert stay.max() < 500.0 and stay[:n_imu].min() > 1.0                                             # an IMU role is microseconds of fp64 chains
    assert tin[:n_imu + 3].max() - tin.min() < 500                                                     # the head of the grid is dispatched at once (5 us)
    chain = n_imu + 2
    assert stay[chain] > stay[:n_imu].max()                                                            # the chain workgroup waits for the IMU roles and then eliminates
    # disarm: a further solve records nothing
    assert be.lib.vil_profile_workgroups(be.ctx, -1, None, 0) == 0
    be.reset_state(); be.solve_resident(opts)
    assert be.lib.vil_profile_workgroups(be.ctx, -1, buf, 4096) == 0
    assert not np.array(buf[:], dtype=np.uint64).any()
    be.close()

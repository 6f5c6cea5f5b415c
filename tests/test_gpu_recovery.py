"""The one-launch iteration's way out: a wait inside the launch that gives up must not end the caller's solve with an error.

The roles of k_iter wait for one another's flags; every wait is bounded by the device's wall clock (50 ms).  When one gives up, vil_solve_resident restores the
state the solve started from and re-runs the SAME solve with two launches per iteration (mode 3 of vil_debug_set_launch_mode), returns that result, and the next
solve is a one-launch solve again.  vil_debug_drop_flag stands in for a workgroup that never became resident."""
import copy
import ctypes as C
import time

import numpy as np
import pytest

from mvil_fusion_amd import abi, lib, synth

pytestmark = pytest.mark.gpu


def bits(s):
    return (s.iterations, s.successful_steps, s.termination, float(s.initial_cost).hex(), float(s.final_cost).hex())


def state_of(be, w):
    be.download_state(w)
    return np.concatenate([w.pose.ravel(), w.speedbias.ravel(), w.ex_pose.ravel(), w.td.ravel(), w.inv_depth.ravel()]).copy()


def counts(be):
    a, b = C.c_int64(0), C.c_int64(0)
    assert be.lib.vil_recovery_counts(be.ctx, C.byref(a), C.byref(b)) == 0
    return a.value, b.value


@pytest.fixture(scope="module")
def window(oracle):
    return synth.make_config(2, L=150, n_plane=3000, n_edge=800, prior_fn=lambda pre: oracle.marginalize(pre).to_prior())


def one_launch(be):
    n, one = C.c_int32(0), C.c_int32(0)
    assert be.lib.vil_debug_get_launch_structure(be.ctx, C.byref(n), C.byref(one)) == 0
    return one.value == 1


@pytest.mark.parametrize("role", ["imu0", "prior", "visual0", "plane0", "gather0"])
def test_lost_flag_is_recovered_by_the_two_launch_structure(oracle, window, role):
    w = window
    be = lib.open_vilsolve()
    be3 = lib.open_vilsolve()
    try:
        assert be3.lib.vil_debug_set_launch_mode(be3.ctx, 3) == 0
        be3.upload(w); be3.reset_state()
        ref3 = bits(be3.solve_resident()); x3 = state_of(be3, copy.deepcopy(w))
        be.upload(w)
        if not one_launch(be):
            pytest.skip("this device does not take the one-launch iteration for the test window")
        be.reset_state(); ref1 = bits(be.solve_resident()); x1 = state_of(be, copy.deepcopy(w))
        assert ref1[:3] == ref3[:3]
        nimu = len(w.imu_i)
        if role == "plane0":
            from test_visual_plan import plan                      # the sweep grid is [imu | prior | rel | visual chunks | plane | edge]
            r = nimu + 2 + plan(w)[0].n_chunks
        else:
            r = {"imu0": 0, "prior": nimu, "visual0": nimu + 2, "gather0": -2}[role]
        before = counts(be)
        assert be.lib.vil_debug_drop_flag(be.ctx, C.c_int32(r), C.c_int32(1)) == 0
        be.reset_state()
        t0 = time.perf_counter()
        s = be.solve_resident()                                    # no error: the solve was re-run
        dt = time.perf_counter() - t0
        assert bits(s) == ref3, (bits(s), ref3)                   # ... with the two-launch structure: that structure's bits
        assert np.array_equal(state_of(be, copy.deepcopy(w)), x3)
        assert 0.04 < dt < 3.0, dt                                  # one bounded wait (50 ms), not a hang and not the host's 2 s poll window twice
        after = counts(be)
        assert after[0] == before[0] + 1 and after[1] == before[1]
        # the abort word is cleared and the next solve is a one-launch solve again: its bits, at its speed
        be.reset_state()
        t0 = time.perf_counter(); s = be.solve_resident(); dt = time.perf_counter() - t0
        assert bits(s) == ref1 and np.array_equal(state_of(be, copy.deepcopy(w)), x1)
        assert dt < 0.1, dt
        assert one_launch(be) and counts(be) == after
        # and it agrees with the oracle like every other solve
        w2 = copy.deepcopy(w); so = oracle.solve(w2)
        assert so.iterations == s.iterations and abs(so.final_cost - s.final_cost) <= 1e-9 * abs(so.final_cost)
    finally:
        be.close(); be3.close()


def test_a_solve_that_fails_on_both_structures_leaves_the_state_unchanged(window):
    w = window
    be = lib.open_vilsolve()
    try:
        be.upload(w)
        if not one_launch(be):
            pytest.skip("this device does not take the one-launch iteration for the test window")
        be.reset_state(); ref1 = bits(be.solve_resident()); x1 = state_of(be, copy.deepcopy(w))
        be.reset_state(); x0 = state_of(be, copy.deepcopy(w))
        assert not np.array_equal(x0, x1)
        assert be.lib.vil_debug_drop_flag(be.ctx, C.c_int32(-2), C.c_int32(2 | 0x10000)) == 0      # gather workgroup 0 loses its flag in launch 2 of both attempts
        before = counts(be)
        with pytest.raises(lib.VilError) as e:
            be.solve_resident()
        assert e.value.status == -2
        assert counts(be) == (before[0], before[1] + 1)
        assert np.array_equal(state_of(be, copy.deepcopy(w)), x0)          # SURVEY 8b "Errors": state left as the solve found it (two accepted steps were undone)
        s = be.solve_resident()                                     # the context is usable: the next solve is an ordinary one-launch solve from that state
        assert bits(s) == ref1 and np.array_equal(state_of(be, copy.deepcopy(w)), x1)
        # vil_solve (upload + solve + download): the caller's arrays are untouched by a solve that fails
        wa = copy.deepcopy(w); keep = wa.pose.copy()
        assert be.lib.vil_debug_drop_flag(be.ctx, C.c_int32(-2), C.c_int32(0 | 0x10000)) == 0
        with pytest.raises(lib.VilError):
            be.solve(wa)
        assert np.array_equal(wa.pose, keep)
        s = be.solve(wa)
        assert bits(s) == ref1
    finally:
        be.close()

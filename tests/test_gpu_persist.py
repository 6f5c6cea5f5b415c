"""The persistent solve (k_solve, csrc/vil_iter.hpp): windows whose every role fits the device at once -- BASELINE configs[1] -- run the whole trust-region solve as ONE
resident launch.  Same trajectory as the oracle and as one launch per iteration (k_iter, vil_debug_set_launch_mode(4)); bit-reproducible; the time cap is read on the device
where ceres reads its clock; a wait that gives up falls back to k_iter."""
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
    w = copy.deepcopy(w)
    be.download_state(w)
    return np.concatenate([w.pose.ravel(), w.speedbias.ravel(), w.ex_pose.ravel(), w.td.ravel(), w.inv_depth.ravel()])


def structure(be):
    n, one = C.c_int32(-1), C.c_int32(0)
    assert be.lib.vil_debug_get_launch_structure(be.ctx, C.byref(n), C.byref(one)) == 0
    return n.value


def counts(be):
    a, b = C.c_int64(0), C.c_int64(0)
    assert be.lib.vil_recovery_counts(be.ctx, C.byref(a), C.byref(b)) == 0
    return a.value, b.value


@pytest.fixture(scope="module")
def w2(oracle):
    return synth.make_config(2, prior_fn=lambda pre: oracle.marginalize(pre).to_prior())


@pytest.fixture(scope="module")
def pair(w2):
    be, be4 = lib.open_vilsolve(), lib.open_vilsolve()
    assert be4.lib.vil_debug_set_launch_mode(be4.ctx, 4) == 0
    be.upload(w2); be4.upload(w2)
    if structure(be) != 0:
        be.close(); be4.close()
        pytest.skip("this device does not hold configs[1]'s roles at once: no persistent solve")
    assert structure(be4) == 1
    yield be, be4
    be.close(); be4.close()


def test_persistent_solve_matches_oracle_and_one_launch_per_iteration(oracle, w2, pair):
    be, be4 = pair
    be.reset_state(); sp = be.solve_resident(); xp = state_of(be, w2)
    be4.reset_state(); s4 = be4.solve_resident(); x4 = state_of(be4, w2)
    wo = copy.deepcopy(w2); so = oracle.solve(wo)
    assert (sp.iterations, sp.successful_steps, sp.termination) == (so.iterations, so.successful_steps, so.termination) == (s4.iterations, s4.successful_steps, s4.termination)
    assert sp.initial_cost == s4.initial_cost                       # the first sweep: the same roles, the same sums
    assert abs(sp.final_cost - so.final_cost) <= 1e-9 * so.final_cost and abs(sp.final_cost - s4.final_cost) <= 1e-12 * s4.final_cost
    for i in range(so.iterations):
        assert abs(sp.cost_trace[i] - so.cost_trace[i]) <= 1e-8 * abs(so.cost_trace[i]) and abs(sp.radius_trace[i] - so.radius_trace[i]) <= 1e-6 * so.radius_trace[i]      # (fp64 tolerances of test_gpu_fullsize.py)
    assert np.abs(xp - x4).max() <= 1e-9
    xo = np.concatenate([wo.pose.ravel(), wo.speedbias.ravel(), wo.ex_pose.ravel(), wo.td.ravel(), wo.inv_depth.ravel()])
    assert np.abs(xp[:7 * w2.K].reshape(-1, 7)[:, :3] - xo[:7 * w2.K].reshape(-1, 7)[:, :3]).max() <= 1e-6
    ref = bits(sp)
    for _ in range(3):                                              # bit-reproducible, solve after solve
        be.reset_state(); s = be.solve_resident()
        assert bits(s) == ref and np.array_equal(state_of(be, w2), xp)
    s = be.solve_resident()                                         # a solve that starts converged: the first judgement of a step ends it (cost first)
    assert s.termination == abi.TERM_NAMES.index("function_tolerance") and s.iterations <= 2


@pytest.mark.parametrize("cap", [1, 3, 5])
def test_iteration_cap_ends_the_persistent_solve_where_the_oracle_ends(oracle, w2, pair, cap):
    be, be4 = pair
    opts = abi.default_options(max_iterations=cap)
    be.reset_state(); s = be.solve_resident(opts)
    be4.reset_state(); s4 = be4.solve_resident(opts)
    wo = copy.deepcopy(w2); so = oracle.solve(wo, opts)
    assert (s.iterations, s.successful_steps, s.termination) == (so.iterations, so.successful_steps, so.termination) == (s4.iterations, s4.successful_steps, s4.termination)
    assert s.termination == abi.TERM_NAMES.index("max_iterations")
    assert abs(s.final_cost - so.final_cost) <= 1e-9 * so.final_cost
    assert np.abs(state_of(be, w2) - state_of(be4, w2)).max() <= 1e-9


def test_time_cap_is_read_on_the_device_where_ceres_reads_its_clock(oracle, w2, pair):
    be, _ = pair
    be.reset_state(); x0 = state_of(be, w2)
    s_full = be.solve_resident()
    # a cap that has expired before the first step: the oracle (clock read at the top of every iteration) returns iteration 0 -- so does the persistent solve
    be.reset_state(); s = be.solve_resident(abi.default_options(max_time_s=1e-7))
    wo = copy.deepcopy(w2); so = oracle.solve(wo, abi.default_options(max_time_s=1e-7))
    assert s.termination == so.termination == abi.TERM_NAMES.index("max_time")
    assert s.iterations == so.iterations == 0 and s.final_cost == s.initial_cost
    assert np.array_equal(state_of(be, w2), x0)
    # a cap in the middle of the solve: an iterate of the un-capped trajectory, reached in about that time
    be.reset_state()
    t0 = time.perf_counter(); s = be.solve_resident(abi.default_options(max_time_s=250e-6)); dt = time.perf_counter() - t0
    assert s.termination == abi.TERM_NAMES.index("max_time") and 1 <= s.iterations < s_full.iterations, (s.termination, s.iterations)
    tr = np.array([s_full.initial_cost] + list(s_full.cost_trace)[:s_full.iterations])
    assert np.abs(tr - s.final_cost).min() <= 1e-12 * s.final_cost
    assert dt < 0.01
    be.reset_state(); assert bits(be.solve_resident()) == bits(s_full)      # the next solve is un-capped again


@pytest.mark.parametrize("role", ["imu0", "visual0", "gather_duty_of_a_tile_workgroup", "gather_duty_of_a_sweep_role"])
def test_a_lost_flag_in_the_persistent_solve_falls_back_to_one_launch_per_iteration(w2, pair, role):
    be, be4 = pair
    be4.reset_state(); ref4 = bits(be4.solve_resident()); x4 = state_of(be4, w2)
    be.reset_state(); refp = bits(be.solve_resident()); xp = state_of(be, w2)
    nimu = len(w2.imu_i)
    r = {"imu0": 0, "visual0": nimu + 2, "gather_duty_of_a_tile_workgroup": -2, "gather_duty_of_a_sweep_role": -2 - 100}[role]
    before = counts(be)
    assert be.lib.vil_debug_drop_flag(be.ctx, C.c_int32(r), C.c_int32(2)) == 0
    be.reset_state()
    t0 = time.perf_counter(); s = be.solve_resident(); dt = time.perf_counter() - t0
    assert bits(s) == ref4 and np.array_equal(state_of(be, w2), x4)      # re-run as one launch per iteration: that structure's bits
    assert 0.04 < dt < 3.0, dt
    assert counts(be) == (before[0] + 1, before[1])
    be.reset_state()
    t0 = time.perf_counter(); s = be.solve_resident(); dt = time.perf_counter() - t0
    assert bits(s) == refp and np.array_equal(state_of(be, w2), xp) and dt < 0.1      # and the next solve is a persistent solve again
    assert structure(be) == 0

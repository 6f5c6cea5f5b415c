"""vil_solve_batch (include/vilsolve.h): B resident windows solved concurrently on one device -- a stream and a host thread per context, one launch per iteration each.
Every window of a batch returns the bits of its solo solve (one launch per iteration, vil_debug_set_launch_mode(4)); different windows in one batch; argument errors."""
import copy
import ctypes as C

import numpy as np
import pytest

from mvil_fusion_amd import abi, lib, synth

pytestmark = pytest.mark.gpu


def bits(be, w, s):
    ww = copy.deepcopy(w); be.download_state(ww)
    return (s.iterations, s.successful_steps, s.termination, float(s.initial_cost).hex(), float(s.final_cost).hex(), ww.pose.tobytes(), ww.speedbias.tobytes(), ww.inv_depth.tobytes())


def solve_batch(bes, opts=None):
    opts = opts or abi.default_options()
    n = len(bes)
    ctxs = (C.c_void_p * n)(*[b.ctx for b in bes])
    sums = (abi.VilSummary * n)(); st = (C.c_int32 * n)()
    f = bes[0].lib.vil_solve_batch; f.restype = C.c_int
    rc = f(ctxs, C.c_int32(n), C.byref(opts), sums, st)
    return rc, list(sums), list(st)


def test_each_window_of_a_batch_returns_the_bits_of_its_solo_solve(oracle):
    pf = lambda pre: oracle.marginalize(pre).to_prior()
    windows = [synth.make_config(2, prior_fn=pf), synth.make_config(2, prior_fn=pf, seed_offset=7), synth.make_config(4, prior_fn=pf), synth.make_config(2, L=300, n_plane=5000, n_edge=1000, prior_fn=pf, seed_offset=3)]
    solo = []
    for w in windows:
        be = lib.open_vilsolve(); assert be.lib.vil_debug_set_launch_mode(be.ctx, 4) == 0
        be.upload(w); be.reset_state(); s = be.solve_resident(); solo.append(bits(be, w, s)); be.close()
    bes = [lib.open_vilsolve() for _ in windows]
    for be, w in zip(bes, windows):
        be.upload(w)
    for rep in range(3):
        for be in bes: be.reset_state()
        rc, sums, st = solve_batch(bes)
        assert rc == 0 and all(v == 0 for v in st)
        for be, w, s, ref in zip(bes, windows, sums, solo):
            assert bits(be, w, s) == ref
    # and against the oracle, like every other solve
    wo = copy.deepcopy(windows[1]); so = oracle.solve(wo)
    assert (sums[1].iterations, sums[1].termination) == (so.iterations, so.termination) and abs(sums[1].final_cost - so.final_cost) <= 1e-8 * so.final_cost
    # a solo solve of the same context afterwards takes the library's own structure again
    bes[0].reset_state(); s = bes[0].solve_resident(); assert s.iterations == solo[0][0]
    # eight at once (two contexts per window) and argument errors
    more = [lib.open_vilsolve() for _ in windows]
    for be, w in zip(more, windows): be.upload(w)
    for be in bes + more: be.reset_state()
    rc, sums, st = solve_batch(bes + more)
    assert rc == 0
    for be, w, s, ref in zip(bes + more, windows + windows, sums, solo + solo):
        assert bits(be, w, s) == ref
    rc, _, _ = solve_batch([bes[0], bes[0]]); assert rc == -1          # the same context twice
    fresh = lib.open_vilsolve(); rc, _, _ = solve_batch([bes[0], fresh]); assert rc == -1      # nothing resident
    for be in bes + more + [fresh]: be.close()

"""Two PROCESSES on one device: the reference runs the estimator and the scan-to-map registration as separate nodes.  The library's gate that
keeps two of its spinning kernels apart is process-local; across processes the guarantee is the bounded wait + the launch-structure ladder (vil_solve_resident).
Here: window solves in a loop in this process beside vmap_align loops in another one -- no error on either side, the peer's results bit-equal to its solo run, every
solve's result bit-equal to ONE of the launch structures' solo results (a solve that lost the device to the peer's kernels mid-launch is re-run one rung down)."""
import copy
import ctypes as C
import os
import subprocess
import sys
import tempfile
import time

import numpy as np
import pytest

from mvil_fusion_amd import abi, lib, synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def bits(be, w, s):
    ww = copy.deepcopy(w); be.download_state(ww)
    return (s.iterations, s.termination, float(s.final_cost).hex(), hash(ww.pose.tobytes() + ww.speedbias.tobytes() + ww.inv_depth.tobytes()))


def run_peer(seconds):
    d = tempfile.mkdtemp(); ready = os.path.join(d, "ready")
    p = subprocess.Popen([sys.executable, os.path.join(HERE, "two_process_peer.py"), str(seconds), ready], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    t0 = time.time()
    while not os.path.exists(ready):
        assert p.poll() is None, p.stderr.read()[-2000:]
        assert time.time() - t0 < 120
        time.sleep(0.05)
    return p


def test_window_solves_beside_scan_to_map_registration_in_another_process(oracle):
    w = synth.make_config(2, prior_fn=lambda pre: oracle.marginalize(pre).to_prior())
    allowed = set()
    for mode in (0, 4, 3):                                # solo results of the three rungs of the ladder
        be = lib.open_vilsolve()
        if mode: assert be.lib.vil_debug_set_launch_mode(be.ctx, mode) == 0
        be.upload(w); be.reset_state(); s = be.solve_resident(); allowed.add(bits(be, w, s)); be.close()
    p = run_peer(1.0); out, err = p.communicate(timeout=120)
    assert p.returncode == 0, err[-2000:]
    solo_n, solo_digest = out.split()[1], out.split()[2]
    assert "," not in solo_digest                         # the peer is deterministic on its own
    be = lib.open_vilsolve(); be.upload(w)
    p = run_peer(3.0)
    n, seen, t_end = 0, set(), time.time() + 2.5
    while time.time() < t_end:                            # no exception = no error status from any solve
        be.reset_state(); s = be.solve_resident(); seen.add(bits(be, w, s)); n += 1
    out, err = p.communicate(timeout=120)
    assert p.returncode == 0, err[-2000:]
    rec, fail = C.c_int64(0), C.c_int64(0)
    be.lib.vil_recovery_counts(be.ctx, C.byref(rec), C.byref(fail))
    print("[two processes on one device] %d window solves beside %s alignments (solo: %s in 1 s); solves re-run one rung down: %d, failed: %d; distinct results %d" % (n, out.split()[1], solo_n, rec.value, fail.value, len(seen)))
    assert n > 100 and fail.value == 0
    assert seen <= allowed, (seen - allowed)
    assert out.split()[2] == solo_digest                  # the peer's results: bit-equal to its solo run
    be.close()

"""The polled completion (pinned host record + sequence word instead of copy + hipStreamSynchronize: include/vilsolve.h, vilvgicp.h, vilmap.h)
and its switch VIL_NO_POLL=1 must give the same results: the same window / scan pair / registration in a child process with the switch
set, compared with this process (bit for bit where the path has no atomics)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from mvil_fusion_amd import abi, lib, synth, vgicp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, sys
sys.path.insert(0, %r)
import __graft_entry__ as g; g.load_package()
import numpy as np
from mvil_fusion_amd import abi, lib, synth, vgicp
be = lib.open_vilsolve()
w = synth.make_config(1)
be.upload(w)
out = []
for _ in range(3):
    be.reset_state(); s = be.solve_resident(abi.default_options())
    out.append([s.iterations, s.termination, s.final_cost.hex()])
tx, tc, sx, sc, T_true = vgicp.make_pair(seed=5, rings=16, az=300)
v = vgicp.Vgicp(lib.load_vilsolve(), "vgicp_"); v.set_target(tx, tc, 0.5); v.set_source(sx, sc)
T, sv = v.align(np.eye(4))
e, H, b, n = v.linearize(np.eye(4))
print(json.dumps({"solve": out, "T": [float(x).hex() for x in T.ravel()], "it": sv.iterations, "e": float(e).hex(), "n": int(n)}))
"""


def run_here():
    be = lib.open_vilsolve()
    w = synth.make_config(1)
    be.upload(w)
    out = []
    for _ in range(3):
        be.reset_state(); s = be.solve_resident(abi.default_options())
        out.append([s.iterations, s.termination, s.final_cost.hex()])
    be.close()
    tx, tc, sx, sc, _ = vgicp.make_pair(seed=5, rings=16, az=300)
    v = vgicp.Vgicp(lib.load_vilsolve(), "vgicp_"); v.set_target(tx, tc, 0.5); v.set_source(sx, sc)
    T, sv = v.align(np.eye(4))
    e, H, b, n = v.linearize(np.eye(4))
    v.close()
    return {"solve": out, "T": [float(x).hex() for x in T.ravel()], "it": sv.iterations, "e": float(e).hex(), "n": int(n)}


def test_polled_and_synchronised_completion_agree():
    assert not os.environ.get("VIL_NO_POLL")
    here = run_here()
    env = dict(os.environ, VIL_NO_POLL="1")
    r = subprocess.run([sys.executable, "-c", CHILD % ROOT], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    there = json.loads(r.stdout.strip().splitlines()[-1])
    if os.environ.get("VIL_POLL_DIAG"):
        with open(os.environ["VIL_POLL_DIAG"], "a") as f: f.write(json.dumps({"here": here, "there": there}) + "\n")
    # What this test guards against is a STALE read -- the host seeing the sequence word before the record: the state or cost of the previous
    # iteration (config 1 ends in a slow valley at the iteration cap: consecutive costs differ by 6e-5 relative), a transform before its last
    # update.  The window solver has no unordered sum left (round 4: the visual role's LDS atomics are gone, every reduction has a fixed order), so
    # two PROCESSES return the same bits; the first VGICP alignment of a cold box differs from later ones in the last bit of one or two entries of T
    # (1.8e-16, every fresh box of four), which keeps that comparison at 1e-12.
    for k in ("it", "e", "n"):
        assert here[k] == there[k], k
    assert max(abs(float.fromhex(a) - float.fromhex(b)) for a, b in zip(here["T"], there["T"])) <= 1e-12
    for a, b in zip(here["solve"] + here["solve"][:1] * 2, there["solve"] + here["solve"][1:]):
        assert a[:2] == b[:2]
        assert a[2] == b[2], (a[2], b[2])                  # the final cost, bit for bit (hex strings)

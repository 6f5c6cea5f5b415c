"""The multi-GPU step (k_step split around the scalar all-reduce, staged all-reduce buffer, RCCL calls
in-stream) exercised on ONE GPU: forced split path, with and without a 1-rank RCCL communicator."""
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
    env = dict(os.environ, VIL_FORCE_SPLIT="1", USE_COMM=use_comm)
    out = subprocess.run([sys.executable, "-c", SCRIPT % (ROOT, ROOT)], env=env, capture_output=True, text=True, timeout=600)
    assert "SPLIT_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]

"""Peer process of tests/test_gpu_two_processes.py: the reference runs scan-to-map registration as a node of its own (lidar_mapping/src/localMapping.cpp:590-791) beside
the estimator node.  This process loops vmap_align -- the persistent k_pose_solve and the association kernels of row f-2 -- on the SAME device for `seconds`, and prints one
line: how many alignments it ran and a digest of every result (the same digest as a solo run = bit-equal results)."""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

g.load_package()
from mvil_fusion_amd import lib, mapreg  # noqa: E402
from mvil_fusion_amd.vgicp import _rot  # noqa: E402

seconds, ready = float(sys.argv[1]), sys.argv[2]
cm, sm = mapreg.make_map(seed=4, n_surf=12000, n_corner=2000)
R, t = _rot(-0.01, 0.015, -0.7), np.array([-2.0, 1.5, 0.2])
sc, ss = mapreg.make_scan(cm, sm, R, t, seed=5, n_surf=2500, n_corner=400)
be = lib.open_vilsolve()
m = mapreg.MapReg(lib.load_vilsolve(), "vmap_")
m.set_map(cm, sm)
q0 = mapreg.quat_from_R(R @ _rot(0.004, -0.003, 0.008)); t0 = t + np.array([0.05, -0.04, 0.03])
digests = set()
m.align(be.ctx, sc, ss, q0, t0)
open(ready, "w").write("ready")                    # the other process starts its loop now
n, t_end = 0, time.time() + seconds
while time.time() < t_end:
    qg, tg, s = m.align(be.ctx, sc, ss, q0, t0)
    digests.add(hashlib.sha256(qg.tobytes() + tg.tobytes() + np.float64(s.final_cost).tobytes()).hexdigest())
    n += 1
print("peer %d %s" % (n, ",".join(sorted(digests))))
m.close(); be.close()

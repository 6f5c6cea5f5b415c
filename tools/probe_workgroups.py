"""When every workgroup of ONE launch of a one-launch iteration (k_iter) entered and left: vil_profile_workgroups.  The phase stamps (tools/probe_phases.py) say when
a role finished; this says when its workgroups were DISPATCHED and how long each stayed -- rounds of workgroups behind the launch's LDS footprint, a role that
queues behind another, a straggler.  CFG=2,3,4 LAUNCH=4 python tools/probe_workgroups.py"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g; g.load_package()
from mvil_fusion_amd import abi, lib, synth
launch = int(os.environ.get("LAUNCH", "4"))
for cfg in [int(v) for v in os.environ.get("CFG", "2").split(",")]:
    be = lib.open_vilsolve()
    w = synth.make_config(cfg); be.upload(w)
    opts = abi.default_options()
    for _ in range(3): be.reset_state(); be.solve_resident(opts)
    be.lib.vil_profile_enable(be.ctx, 1)
    be.lib.vil_profile_workgroups.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_uint64), C.c_int32]
    assert be.lib.vil_profile_workgroups(be.ctx, launch, None, 0) == 0
    be.reset_state(); s = be.solve_resident(opts)
    buf = (C.c_uint64 * (2 * 4096))()
    assert be.lib.vil_profile_workgroups(be.ctx, launch, buf, 4096) == 0
    a = np.array(buf[:], dtype=np.uint64).reshape(-1, 2)
    idx = np.nonzero(a[:, 0])[0]
    if len(idx) == 0: print("cfg %d: nothing recorded (not a one-launch iteration, or fewer than %d launches)" % (cfg, launch + 1)); be.close(); continue
    t0 = a[idx, 0].min(); tin = (a[idx, 0] - t0) * 0.01; tout = (a[idx, 1] - t0) * 0.01
    print("cfg %d, launch %d of %d iterations: %d workgroups; us after the first one entered" % (cfg, launch, s.iterations, len(idx)))
    def grp(name, sel):
        if sel.sum(): print("  %-12s n %4d | entered %6.2f .. %6.2f (median %6.2f) | left %6.2f .. %6.2f (median %6.2f) | stayed median %5.2f max %5.2f" % (name, sel.sum(), tin[sel].min(), tin[sel].max(), np.median(tin[sel]), tout[sel].min(), tout[sel].max(), np.median(tout[sel]), np.median(tout[sel] - tin[sel]), (tout[sel] - tin[sel]).max()))
    n_imu = w.K - 1
    grp("imu", idx < n_imu); grp("prior", idx == n_imu); grp("icp/lps", idx == n_imu + 1); grp("chain", idx == n_imu + 2)
    for lo in range(n_imu + 3, int(idx.max()) + 1, 64): grp("wg %d.." % lo, (idx >= lo) & (idx < lo + 64))
    be.close()

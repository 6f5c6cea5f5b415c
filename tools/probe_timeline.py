"""Per-iteration timeline of one solve from the kernels' own wall-clock stamps (vil_profile_enable 2 = stamps alone: the persistent solve keeps its launch).
CFG=2 MODE=0 python tools/probe_timeline.py   -> one row per iteration, microseconds since the solve's first stamp"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g; g.load_package()
from mvil_fusion_amd import abi, lib, synth
SHOW = [(0, "start"), (17, "imu staged"), (20, "imu rec"), (2, "imu done"), (3, "ch seen"), (14, "ch slab"), (1, "sweep done"), (6, "g saw vis"), (5, "g done"), (8, "m saw g"), (4, "W^T"), (9, "m tiles"),
        (10, "chol"), (11, "x_p"), (21, "ch back"), (22, "sums in"), (23, "cand"), (12, "m done"), (24, "tail in"), (25, "hdr out"), (26, "ctl out"), (27, "r0 hdr seen"), (28, "r0 next")]
cfg = int(os.environ.get("CFG", "2")); mode = int(os.environ.get("MODE", "0"))
be = lib.open_vilsolve()
if mode: assert be.lib.vil_debug_set_launch_mode(be.ctx, mode) == 0
w = synth.make_config(cfg); be.upload(w)
TOL = os.environ.get("TOLERATE") is not None      # (timing experiments with wrong results: a failing solve still leaves its stamps)
def solve():
    try: return be.solve_resident()
    except lib.VilError as e:
        if not TOL: raise
        return None
for _ in range(3): be.reset_state(); solve()
lpi_, one_ = C.c_int32(0), C.c_int32(0); be.lib.vil_debug_get_launch_structure(be.ctx, C.byref(lpi_), C.byref(one_))
be.lib.vil_profile_enable(be.ctx, int(os.environ.get("PROF", "2" if lpi_.value == 0 else "1")))
be.reset_state(); s = solve()
its = s.iterations if s is not None else 2
buf = (C.c_uint64 * (32 * 64))()
n = be.lib.vil_debug_read_stamps(be.ctx, buf, 64)
M = 0xFFFFFFFFFFFFFFFF
t00 = None
print("iterations %d; columns: " % its + " | ".join(nm for _, nm in SHOW))
for q in range(min(n, its + 2)):
    r = buf[32 * q: 32 * q + 32]
    if not r[0]: continue
    t0 = (~r[0]) & M
    if t00 is None: t00 = t0
    row = ["%6.1f" % ((t0 - t00) * 0.01)]
    for k, nm in SHOW[1:]: row.append("%5.1f" % ((r[k] - t0) * 0.01) if r[k] >= t0 else "    -")
    print("it %2d abs %s | rel: %s" % (q, row[0], " ".join(row[1:])))

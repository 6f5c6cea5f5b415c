"""Where a one-launch iteration (k_iter) spends its time: average position of the launch's own wall-clock stamps (vil_profile_phases), per BASELINE window.
CFG=2,3,4 python tools/probe_phases.py   (MODE=3: the two-launch structure, for the it/s comparison only)"""
import sys, os, ctypes as C, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g; g.load_package()
from mvil_fusion_amd import abi, lib, synth
NAMES = ["first wg start", "last visual/lidar/rel role done", "last imu role done", "chain: records seen", "chain: W^T complete", "last gather wg done", "last gather wg saw visual flags", "master started",
         "master saw gather flags", "master saw tiles", "cholesky done", "x_p published", "master done", "last tile wg done", "chain: slab gathered", "prior role done", "imu0: role entered", "imu0: inputs staged", "imu0: raw blocks done", "imu0: whitened", "imu0: record stores issued", "master: chain back-substituted", "master: step vectors + helpers' sums in", "master: candidate formed", "tail: hflag2 + candidate stores in", "tail: header posted", "tail: Ctl out, word 8 posted", "sweep role 0: header seen", "sweep role 0: next iteration entered", "", "", ""]
for cfg in [int(v) for v in os.environ.get("CFG", "2").split(",")]:
    be = lib.open_vilsolve()
    mode = int(os.environ.get("MODE", "0"))
    if mode: assert be.lib.vil_debug_set_launch_mode(be.ctx, mode) == 0
    w = synth.make_config(cfg)
    be.upload(w)
    opts = abi.default_options()
    for _ in range(3): be.reset_state(); be.solve_resident(opts)
    t0 = time.perf_counter(); its = 0
    for _ in range(20): be.reset_state(); its += be.solve_resident(opts).iterations
    el = time.perf_counter() - t0
    print("cfg %d mode %d: %.0f it/s (%.1f us per iteration, host clock, graph replay)" % (cfg, mode, its / el, 1e6 * el / its))
    if mode in (0, 4):
        lpi_, one_ = C.c_int32(0), C.c_int32(0); be.lib.vil_debug_get_launch_structure(be.ctx, C.byref(lpi_), C.byref(one_))
        be.lib.vil_profile_enable(be.ctx, int(os.environ.get("PROF", "2" if lpi_.value == 0 else "1")))      # 2: stamps alone (the persistent solve keeps its launch)
        for _ in range(5): be.reset_state(); be.solve_resident(opts)
        avg = (C.c_double * 32)(); n = C.c_int64(0)
        be.lib.vil_profile_phases(be.ctx, avg, C.byref(n), 1)
        be.lib.vil_profile_enable(be.ctx, 0)
        print("  %d launches averaged; iteration period inside a persistent solve %.2f us; us after the launch's / iteration's first workgroup started:" % (n.value, avg[0]))
        for k in sorted([q for q in range(1, 32) if avg[q] > 0], key=lambda q: avg[q]): print("    %6.2f  %s" % (avg[k], NAMES[k]))
    be.close()

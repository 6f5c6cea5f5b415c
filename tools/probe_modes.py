"""Debug aid: one configs[cid] solve per launch structure, each in its own process (a device fault ends that process only).  usage: CFG=2 python tools/probe_modes.py [modes...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    import ctypes as C
    import __graft_entry__ as g; g.load_package()
    from mvil_fusion_amd import abi, lib, synth
    mode, cid = int(sys.argv[2]), int(sys.argv[3])
    kw = {} if len(sys.argv) < 5 else eval(sys.argv[4])
    be = lib.open_vilsolve()
    if mode: assert be.lib.vil_debug_set_launch_mode(be.ctx, mode) == 0
    w = synth.make_config(cid, **kw)
    be.upload(w)
    n, one = C.c_int32(0), C.c_int32(0)
    be.lib.vil_debug_get_launch_structure(be.ctx, C.byref(n), C.byref(one))
    import time
    for rep in range(3):
        be.reset_state(); t0 = time.perf_counter(); s = be.solve_resident(); dt = time.perf_counter() - t0
        print("mode %d cfg %d launches/iter %d: it %d succ %d term %d cost %.12g -> %.15g  %.3f ms" % (mode, cid, n.value, s.iterations, s.successful_steps, s.termination, s.initial_cost, s.final_cost, dt * 1e3), float(s.final_cost).hex(), flush=True)
    sys.exit(0)
cid = int(os.environ.get("CFG", "2"))
extra = os.environ.get("KW", "{}")
for mode in (sys.argv[1:] or ["4", "0"]):
    r = subprocess.run(["timeout", "60", sys.executable, __file__, "--one", mode, str(cid), extra], capture_output=True, text=True)
    print(r.stdout.strip()); 
    if r.returncode != 0: print("mode %s: rc %d\n%s" % (mode, r.returncode, r.stderr[-1500:]))

"""Phase timing of the step kernel: build csrc with -DVIL_STAMPS first (s_memtime stamps land in P.dbg), then run this on a GPU box."""
import sys, os, ctypes as C, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g; g.load_package()
from mvil_fusion_amd import abi, lib, synth
import numpy as np
from bench import VilProfile
cfg = int(os.environ.get("CFG", "2"))
be = lib.open_vilsolve()
w = synth.make_config(cfg)
be.upload(w)
be.lib.vil_profile_enable(be.ctx, 1)
opts = abi.default_options()
def _solve():
    try: return be.solve_resident(opts)
    except Exception as e: return None
for _ in range(3): be.reset_state(); _solve()
prof = VilProfile(); be.lib.vil_profile_read(be.ctx, C.byref(prof), 1)
t0=time.perf_counter(); its=0
for _ in range(20):
    be.reset_state(); s=_solve(); its+=(s.iterations if s else 1)
if s is None: s=type('S',(),{'iterations':-1})()
el=time.perf_counter()-t0
be.lib.vil_profile_read(be.ctx, C.byref(prof), 1)
print("cfg", cfg, "skip", os.environ.get("VIL_SKIP"), "vwg", os.environ.get("VIL_VWG"), "it/s %.0f"%(its/el), "iters", s.iterations, "sweep_us %.1f"%(1e3*prof.sweep_ms/max(1,prof.sweep_launches)), "reduce+step_us %.1f"%(1e3*prof.step_ms/max(1,prof.step_launches)), "reduce_us %.1f"%(1e3*prof.reduce_ms/max(1,prof.step_launches)))
dbg = (C.c_longlong*64)(); be.lib.vil_debug_read(be.ctx, dbg)
d = np.array(dbg[:12], dtype=np.int64); print("raw", (d-d[0]).tolist()); print("backsubst phase I end stamp rel to stamp3:", dbg[23]-dbg[3], "of total", dbg[4]-dbg[3])
print("step stamps (cycles, deltas):", np.diff(d).tolist())
v = np.array(dbg[32:40], dtype=np.int64); print("visual WG0 stamps rel:", (v - v[0]).tolist())
print("cholesky cycles (wave 0): diag %d, panel %d, barrier after panel %d, trailing %d, barrier after trailing %d, publish+barrier %d" % (dbg[20], dbg[26], dbg[21], dbg[22], dbg[24], dbg[25]))
print("packing loop, wave 0 own cycles:", dbg[27], " stamps 10->11:", dbg[11]-dbg[10], " 9->10:", dbg[10]-dbg[9], " 1->9:", dbg[9]-dbg[1], " 11->2:", dbg[2]-dbg[11], " 4->5:", dbg[5]-dbg[4], " 5->6:", dbg[6]-dbg[5])
c = np.array(dbg[40:47], dtype=np.int64)
print("solve path stamps 40..46 (ticks, deltas): entry -> [1] tiles/chain ready -> [2],[3] packed -> [4] dense cholesky -> [5] back-subst -> [6] chain back-subst:", np.diff(c).tolist())
print("merged launch: kernel entry -> gather flags seen + Ctl loaded (stamp 0): %d ticks; 1 -> 9 (vectors): %d; 9 -> 10: %d; 10 -> solve entry: %d" % (dbg[0] - dbg[30], dbg[9] - dbg[1], dbg[10] - dbg[9], dbg[40] - dbg[10]))
f = np.array(dbg[12:17], dtype=np.int64)
print("chain workgroup (ticks): zero + table gather %d, scales %d, chain %d, write-out %d" % tuple(np.diff(f).tolist()))
print("whole master: entry -> final stamp 7: %d ticks = %.1f us" % (dbg[7] - dbg[30], (dbg[7] - dbg[30]) / 2390.0))
g = np.array([dbg[14]] + [dbg[48 + q] for q in range(7)] + [dbg[15]], dtype=np.int64)
print("chain workgroup, forward recursion wave (ticks after the scales): block 0..5 published, middle block published, all waves done:", (g[1:] - g[0]).tolist(), " row waves done: forward %d, backward %d" % (dbg[55] - dbg[14], dbg[56] - dbg[14]))
print("visual WG 0, ticks summed over its chunks: factor evaluation %d, landmark sums %d, outer products %d" % (dbg[62], dbg[63], dbg[47]))
print("visual workgroups: longest %d ticks = %.1f us, sum %d ticks; WG 0 phases (zero+eval | landmark sums + fill | matrix cores + record | cost):" % (dbg[60], dbg[60] / 2390.0, dbg[61]), np.diff(np.array(dbg[32:37], dtype=np.int64)).tolist())
im = np.array([dbg[17], dbg[18], dbg[19], dbg[28]], dtype=np.int64)
print("IMU workgroup 0 (ticks): raw blocks %d, whitening %d, contraction + record %d" % tuple(np.diff(im).tolist()), "| chain workgroup: entry -> IMU / prior flags seen %d, -> gathered %d" % (dbg[31] - dbg[12], dbg[13] - dbg[31]))

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
for _ in range(3): be.reset_state(); be.solve_resident(opts)
prof = VilProfile(); be.lib.vil_profile_read(be.ctx, C.byref(prof), 1)
t0=time.perf_counter(); its=0
for _ in range(20): be.reset_state(); s=be.solve_resident(opts); its+=s.iterations
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
print("chain path (cycles): init %d, chain steps %d, middle %d, schur+dense cholesky %d, dense back-subst %d, chain back-subst %d" % tuple(np.diff(c).tolist()))
e = np.array(dbg[48:59], dtype=np.int64) - dbg[41]
print("chain waves, ticks after the start: recursion steps", e[:6].tolist(), "middle done", int(e[6]), "| fwd row wave done", int(e[7]), "| bwd row wave done", int(e[8]), "| pack done", int(e[9]), "| q done", int(e[10]))
f = np.array(dbg[12:18], dtype=np.int64)
print("chain workgroup in k_sweep (ticks): wait for IMU/prior %d, stage %d, scales %d, chain (wave 0) %d, write-out %d; visual WG0 ends %d ticks after the chain WG started" % (tuple(np.diff(f).tolist()) + (int(dbg[37] - dbg[12]),)))
print("visual workgroups: longest entry-to-exit of any of them in any launch since the upload: %d ticks" % dbg[60])

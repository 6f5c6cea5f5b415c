import sys, os, ctypes as C, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import __graft_entry__ as g; g.load_package()
from mvil_fusion_amd import abi, lib, synth
cfg=int(os.environ.get("CFG","4")); mode=int(os.environ.get("MODE","0"))
be = lib.open_vilsolve()
if mode: be.lib.vil_debug_set_launch_mode(be.ctx, mode)
w = synth.make_config(cfg); be.upload(w); opts = abi.default_options()
ts=[]
for _ in range(12):
    be.reset_state(); t0=time.perf_counter(); s=be.solve_resident(opts); ts.append((time.perf_counter()-t0)*1e6)
print("cfg",cfg,"mode",mode,"graph",os.environ.get("VIL_GRAPH"),"iters",s.iterations,"solve us:",[int(x) for x in ts])

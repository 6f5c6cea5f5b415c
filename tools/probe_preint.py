"""Wall time of vpre_integrate through ctypes: whole wrapper vs the bare C call (pre-built arguments)."""
import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
if os.environ.get("PROBE_TORCH"):
    import torch
import __graft_entry__ as g; g.load_package()
from mvil_fusion_amd import lib, preint
s = preint.make_stream(9, (20, 40), seed=20240608)
p = preint.Preint(lib.load_vilsolve(), "vpre_")
for _ in range(5): p.integrate(*s)
N = 300
t = time.perf_counter()
for _ in range(N): p.integrate(*s)
print("wrapper  %.1f us" % ((time.perf_counter() - t) / N * 1e6))
dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
arrs = [np.ascontiguousarray(x, np.float64) for x in s[1:]] + [preint.NOISE.copy()]
start = np.ascontiguousarray(s[0], np.int32); out = np.zeros((9, 287)); jac = np.zeros((9, 225))
f = p.lib.vpre_integrate; f.restype = C.c_int
args = [p.ctx, C.c_int32(9), start.ctypes.data_as(ip)] + [a.ctypes.data_as(dp) for a in arrs] + [out.ctypes.data_as(dp), jac.ctypes.data_as(dp)]
t = time.perf_counter()
for _ in range(N): f(*args)
print("bare     %.1f us" % ((time.perf_counter() - t) / N * 1e6))
args[-1] = None
t = time.perf_counter()
for _ in range(N): f(*args)
print("bare, no jacobian %.1f us" % ((time.perf_counter() - t) / N * 1e6))

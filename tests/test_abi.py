"""No-GPU checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/vilsolve.h declares; compute entry points refuse to run without a device (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from mvil_fusion_amd import abi, lib, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "vilsolve.h")).read()
    src = src[src.index("/* ---- entry points"):]
    return sorted(set(re.findall(r"\b(vil_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    so = lib.load_vilsolve()
    syms = declared_symbols()
    assert len(syms) >= 18
    for s in syms:
        assert hasattr(so, s), "libvilsolve.so does not export %s" % s
    assert so.vil_abi_version() == 1


def test_library_exports_every_vgicp_symbol():
    """include/vilvgicp.h is served by the same shared library."""
    so = lib.load_vilsolve()
    src = open(os.path.join(ROOT, "include", "vilvgicp.h")).read()
    syms = sorted(set(re.findall(r"\b(vgicp_[a-z_0-9]+)\s*\(", src)))
    assert len(syms) == 12, syms
    for s in syms:
        assert hasattr(so, s), "libvilsolve.so does not export %s" % s
    import subprocess, tempfile
    from mvil_fusion_amd import vgicp
    prog = '#include <stdio.h>\n#include "vilvgicp.h"\nint main(void){printf("%zu %zu\\n", sizeof(vgicp_options), sizeof(vgicp_summary));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(prog)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"), "-o", os.path.join(d, "s")])
        out = subprocess.check_output([os.path.join(d, "s")]).decode().split()
    assert [C.sizeof(vgicp.VgicpOptions), C.sizeof(vgicp.VgicpSummary)] == [int(v) for v in out]
    with pytest.raises(vgicp.VgicpError):
        import torch
        if torch.cuda.is_available():
            raise vgicp.VgicpError("GPU present")
        vgicp.Vgicp(so, "vgicp_")                       # no device -> refuses, no CPU fallback


def test_library_exports_every_vmap_symbol():
    """include/vilmap.h (scan-to-map registration) is served by the same shared library."""
    so = lib.load_vilsolve()
    src = open(os.path.join(ROOT, "include", "vilmap.h")).read()
    syms = sorted(set(re.findall(r"\b(vmap_[a-z_0-9]+)\s*\(", src)))
    assert len(syms) == 8, syms
    for s in syms:
        assert hasattr(so, s), "libvilsolve.so does not export %s" % s
    import subprocess, tempfile
    from mvil_fusion_amd import mapreg
    prog = '#include <stdio.h>\n#include "vilmap.h"\nint main(void){printf("%zu\\n", sizeof(vmap_summary));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(prog)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"), "-o", os.path.join(d, "s")])
        out = subprocess.check_output([os.path.join(d, "s")]).decode().split()
    assert C.sizeof(mapreg.VmapSummary) == int(out[0])
    with pytest.raises(mapreg.MapRegError):
        import torch
        if torch.cuda.is_available():
            raise mapreg.MapRegError("GPU present")
        mapreg.MapReg(so, "vmap_")                      # no device -> refuses, no CPU fallback


def test_library_exports_every_vpre_symbol():
    """include/vilpreint.h (IMU pre-integration) is served by the same shared library."""
    so = lib.load_vilsolve()
    src = open(os.path.join(ROOT, "include", "vilpreint.h")).read()
    syms = sorted(set(re.findall(r"\b(vpre_[a-z_0-9]+)\s*\(", src)))
    assert len(syms) == 5, syms
    for s in syms:
        assert hasattr(so, s), "libvilsolve.so does not export %s" % s
    from mvil_fusion_amd import preint
    with pytest.raises(preint.PreintError):
        import torch
        if torch.cuda.is_available():
            raise preint.PreintError("GPU present")
        preint.Preint(so, "vpre_")                      # no device -> refuses, no CPU fallback


def test_release_build_has_no_tuning_knobs():
    """The development knobs (csrc/vil_tuning.hpp: chunking, kernel variants, role masks, debug prints) exist only in the
    -DVIL_TUNING build; the shipping library does not even contain their names."""
    blob = open(lib.LIB_PATH, "rb").read()
    for knob in (b"VIL_SKIP", b"VIL_HELP", b"VIL_VWG", b"VIL_VFBAL", b"VIL_DENSE_STEP", b"VIL_MARG_PIVOTED", b"VIL_MARG_DEBUG", b"VIL_GRAPH",
                 b"VIL_FORCE_SPLIT", b"VIL_PRECHAIN", b"VIL_NO_MERGE", b"VIL_GATHER32", b"VIL_UPLOAD_TRACE", b"VIL_MAP_FUSED_MAX", b"VGICP_", b"VPRE_TIMING"):
        assert knob not in blob, knob


def test_struct_layouts_match_header():
    """sizeof() of the ctypes mirrors equals what the C compiler lays out (checked through a tiny C program)."""
    import subprocess, tempfile
    prog = r'''
#include <stdio.h>
#include "vilsolve.h"
int main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(vil_state), sizeof(vil_prior), sizeof(vil_problem), sizeof(vil_options), sizeof(vil_summary), sizeof(vil_marg_spec), sizeof(vil_prior_out), sizeof(vil_device_cfg));return 0;}
'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(prog)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"), "-o", os.path.join(d, "s")])
        out = subprocess.check_output([os.path.join(d, "s")]).decode().split()
    got = [C.sizeof(t) for t in (abi.VilState, abi.VilPrior, abi.VilProblem, abi.VilOptions, abi.VilSummary, abi.VilMargSpec, abi.VilPriorOut, abi.VilDeviceCfg)]
    assert got == [int(v) for v in out]


def test_host_helpers_without_device():
    so = lib.load_vilsolve()
    assert so.vil_reduced_dim(10) == 157
    o = abi.VilOptions()
    so.vil_default_options(C.byref(o))
    assert o.max_iterations == 30 and o.visual_loss == abi.LOSS_CAUCHY and abi.f64([o.lidar_loss_scale])[0] == 0.1
    so.vil_strerror.restype = C.c_char_p
    assert b"device" in so.vil_strerror(-2).lower()


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(lib.VilError) as e:
        lib.open_vilsolve()
    assert e.value.status == -2


def test_shard_ranges_partition(oracle):
    so = lib.load_vilsolve()
    w = synth.make_config(2, L=120, n_plane=1000, n_edge=333)
    p = w.c_problem()
    for world in (1, 2, 3, 8):
        got = {k: [] for k in "lep"}
        for r in range(world):
            v = [C.c_int32() for _ in range(6)]
            assert so.vil_shard_ranges(C.byref(p), r, world, *[C.byref(x) for x in v]) == 0
            got["l"].append((v[0].value, v[1].value)); got["e"].append((v[2].value, v[3].value)); got["p"].append((v[4].value, v[5].value))
        for key, n in (("l", w.L), ("e", 333), ("p", 1000)):
            assert got[key][0][0] == 0 and got[key][-1][1] == n
            for a, b in zip(got[key][:-1], got[key][1:]):
                assert a[1] == b[0] and a[0] <= a[1]


def test_gauge_fix_matches_oracle(oracle):
    so = lib.load_vilsolve()
    w = synth.make_config(1)
    p0 = w.pose[0].copy()
    rng = np.random.default_rng(0)
    # move the whole window by a yaw + translation, as an unconstrained solve would
    yaw = 0.3
    Rz = np.array([[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1]])
    for k in range(w.K):
        w.pose[k, :3] = Rz @ w.pose[k, :3] + [1.0, -2.0, 0.5]
        w.pose[k, 3:] = synth.R_to_quat(Rz @ synth.quat_to_R(w.pose[k, 3:]))
        w.speedbias[k, :3] = Rz @ w.speedbias[k, :3]
    wa, wb = w, synth.make_config(1)
    wb.set_state(wa.state_copy())
    s = wa.c_state()
    so.vil_gauge_fix.restype = C.c_int
    assert so.vil_gauge_fix(p0.ctypes.data_as(C.POINTER(C.c_double)), C.byref(s)) == 0
    oracle.gauge_fix(p0, wb)
    assert np.allclose(wa.pose, wb.pose, atol=1e-13) and np.allclose(wa.speedbias, wb.speedbias, atol=1e-13)
    w0 = synth.make_config(1)
    assert np.allclose(wa.pose[:, :3], w0.pose[:, :3], atol=1e-12)   # yaw+translation fully undone

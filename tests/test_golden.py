"""Golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py): inputs + expected outputs.
CPU: the oracle reproduces them (guards the restatement against regressions and compiler / flag drift).
GPU: the HIP path reproduces them through the C-ABI."""
import glob
import os

import numpy as np
import pytest

from mvil_fusion_amd import abi

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = sorted(glob.glob(os.path.join(HERE, "golden", "*.npz")))


def load(path):
    d = np.load(path)
    w = abi.Window.from_dict({k[3:]: d[k] for k in d.files if k.startswith("in_")})
    return d, w


def rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300) if b.size else 0.0


def check_backend(be, path, tol_f, tol_s):
    d, w = load(path)
    for cls in range(7):
        r, J = be.eval_factors(w, cls)
        assert rel(r, d["r%d" % cls]) <= tol_f and rel(J, d["J%d" % cls]) <= tol_f, (os.path.basename(path), cls)
    cost, S, g = be.linearize(w)
    assert abs(cost - d["lin_cost"][0]) <= 1e-12 * d["lin_cost"][0]
    assert rel(S, d["lin_S"]) <= tol_s and rel(g, d["lin_g"]) <= tol_s
    p0 = w.pose[0].copy()
    sm = be.solve(w)
    be.gauge_fix(p0, w)
    exp = d["sol_summary"]
    assert (sm.iterations, sm.successful_steps, sm.termination) == (int(exp[0]), int(exp[1]), int(exp[2]))
    assert abs(sm.final_cost - exp[4]) <= 1e-7 * exp[4]
    assert np.abs(w.pose[:, :3] - d["sol_pose"][:, :3]).max() <= 1e-6
    assert np.abs(w.pose[:, 3:] - d["sol_pose"][:, 3:]).max() <= 1e-7
    assert rel(w.inv_depth, d["sol_inv_depth"]) <= 1e-5
    w2 = abi.Window.from_dict({k[3:]: d[k] for k in d.files if k.startswith("in_")})
    w2.set_state({k: d["sol_" + k] for k in ("pose", "speedbias", "ex_pose", "td", "inv_depth")})
    mo = be.marginalize(w2, abi.MARGIN_OLD)
    assert [mo.c.n, mo.c.nblk, mo.c.m] == [int(v) for v in d["marg_meta"]]
    assert np.array_equal(mo.blk_kind[:mo.c.nblk], d["marg_kind"]) and np.array_equal(mo.blk_index[:mo.c.nblk], d["marg_index"]) and np.array_equal(mo.blk_col[:mo.c.nblk], d["marg_col"])
    assert rel(mo.A_matrix(), d["marg_A"]) <= 1e-8 and rel(mo.b_vector(), d["marg_b"]) <= 1e-8


def test_fixtures_present():
    assert len(FILES) >= 3


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_oracle_reproduces_golden(oracle, path):
    check_backend(oracle, path, 1e-13, 1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_hip_reproduces_golden(hip, path):
    check_backend(hip, path, 1e-12, 1e-10)

"""GPU parity of the scan-to-map registration (include/vilmap.h) against the CPU oracle, through the C-ABI."""
import numpy as np
import pytest

from mvil_fusion_amd import abi, lib, mapreg
from mvil_fusion_amd.vgicp import _rot

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup(oracle):
    cm, sm = mapreg.make_map(seed=4, n_surf=12000, n_corner=2000)
    R, t = _rot(-0.01, 0.015, -0.7), np.array([-2.0, 1.5, 0.2])
    sc, ss = mapreg.make_scan(cm, sm, R, t, seed=5, n_surf=2500, n_corner=400)
    g = mapreg.MapReg(lib.load_vilsolve(), "vmap_"); o = mapreg.MapReg(oracle.lib, "orc_vmap_")
    for r in (g, o):
        r.set_map(cm, sm)
    yield g, o, sc, ss, R, t
    g.close(); o.close()


def test_association_parity(setup):
    g, o, sc, ss, R, t = setup
    q = mapreg.quat_from_R(R @ _rot(0.003, 0.002, -0.006)); t0 = t + np.array([0.04, 0.03, -0.02])
    eg, pg = g.associate(sc, ss, q, t0)
    eo, po = o.associate(sc, ss, q, t0)
    assert eg.shape == eo.shape and pg.shape == po.shape and len(eg) > 100 and len(pg) > 1000
    assert np.array_equal(eg[:, :3], eo[:, :3]) and np.array_equal(pg[:, :3], po[:, :3])         # the same scan points were accepted
    # the direction of a fitted line is defined up to sign: (a, b) may come out swapped
    same = np.abs(eg[:, 3:] - eo[:, 3:]).max(axis=1) < 1e-9
    swapped = np.abs(eg[:, 3:6] - eo[:, 6:9]).max(axis=1) + np.abs(eg[:, 6:9] - eo[:, 3:6]).max(axis=1) < 1e-9
    assert np.all(same | swapped)
    assert np.abs(pg[:, 3:] - po[:, 3:]).max() < 1e-9


def test_align_parity(hip, setup):
    g, o, sc, ss, R, t = setup
    q0 = mapreg.quat_from_R(R @ _rot(0.004, -0.003, 0.008)); t0 = t + np.array([0.05, -0.04, 0.03])
    qg, tg, sg = g.align(hip.ctx, sc, ss, q0, t0)
    qo, to, so = o.align(None, sc, ss, q0, t0)
    assert (sg.rounds, sg.n_edge, sg.n_plane, sg.iterations) == (so.rounds, so.n_edge, so.n_plane, so.iterations)
    assert abs(sg.final_cost - so.final_cost) <= 1e-8 * so.final_cost
    assert np.abs(tg - to).max() < 1e-8 and np.abs(qg - qo).max() < 1e-9
    assert np.linalg.norm(tg - t) < 0.02


def test_empty_inputs(hip, setup):
    g, o, sc, ss, R, t = setup
    q = mapreg.quat_from_R(R)
    e, p = g.associate(sc[:0], ss[:0], q, t)
    assert len(e) == 0 and len(p) == 0
    far = np.array([500.0, 0.0, 0.0])
    e, p = g.associate(sc, ss, q, far)                               # nothing within 1 m of the map
    assert len(e) == 0 and len(p) == 0


def test_gpu_reproduces_golden_fixture(hip):
    from test_oracle_map import _check_golden
    g = mapreg.MapReg(lib.load_vilsolve(), "vmap_")
    _check_golden(g, hip.ctx)
    g.close()

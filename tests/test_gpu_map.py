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


def test_duplicate_map_points_and_ties(oracle):
    """Exact distance ties (duplicated map points, a scan point coinciding with map points): the (distance, index) order of the
    wave search is the oracle's, so the same neighbours -- and the same factors -- come out."""
    cm, sm = mapreg.make_map(seed=8, n_surf=5000, n_corner=900)
    cm2 = np.concatenate([cm, cm[:400], cm[:150]]); sm2 = np.concatenate([sm, sm[:2500]])              # up to three copies of a point
    R, t = _rot(0.0, 0.0, 0.2), np.array([0.5, 0.5, 0.1])
    sc, ss = mapreg.make_scan(cm, sm, R, t, seed=9, n_surf=900, n_corner=200)
    # a few scan points that land exactly on map points (zero distance to all copies)
    ss[:20, :3] = ((sm[:20, :3].astype(np.float64) - t) @ R).astype(np.float32); sc[:10, :3] = ((cm[:10, :3].astype(np.float64) - t) @ R).astype(np.float32)
    g = mapreg.MapReg(lib.load_vilsolve(), "vmap_"); o = mapreg.MapReg(oracle.lib, "orc_vmap_")
    q = mapreg.quat_from_R(R)
    for r in (g, o):
        r.set_map(cm2, sm2)
    eg, pg = g.associate(sc, ss, q, t); eo, po = o.associate(sc, ss, q, t)
    g.close(); o.close()
    assert eg.shape == eo.shape and pg.shape == po.shape and len(pg) > 300
    assert np.array_equal(eg[:, :3], eo[:, :3]) and np.array_equal(pg[:, :3], po[:, :3])
    assert np.abs(pg[:, 3:] - po[:, 3:]).max() < 1e-8


def test_sparse_map_far_queries_and_fallback(oracle):
    """A map so sparse that most queries need several rings or the exhaustive fallback, and a cell size the adaptive build
    shrinks: still the oracle's factors."""
    rng = np.random.default_rng(3)
    sm = np.concatenate([rng.uniform(-40, 40, (600, 2)), np.zeros((600, 1)) + rng.normal(0, 0.01, (600, 1)), rng.uniform(0, 50, (600, 1))], axis=1).astype(np.float32)   # a ground plane, ~0.1 pt / m^2
    cm = np.concatenate([np.linspace(-30, 30, 400)[:, None], np.zeros((400, 1)), np.ones((400, 1)), np.zeros((400, 1))], axis=1).astype(np.float32)                     # one long edge
    ss = np.concatenate([rng.uniform(-35, 35, (300, 2)), rng.normal(0, 0.02, (300, 1)), rng.uniform(0, 50, (300, 1))], axis=1).astype(np.float32)
    sc = np.concatenate([rng.uniform(-25, 25, (80, 1)), rng.normal(0, 0.02, (80, 1)), 1 + rng.normal(0, 0.02, (80, 1)), np.zeros((80, 1))], axis=1).astype(np.float32)
    q, t = np.array([0, 0, 0, 1.0]), np.zeros(3)
    g = mapreg.MapReg(lib.load_vilsolve(), "vmap_"); o = mapreg.MapReg(oracle.lib, "orc_vmap_")
    for r in (g, o):
        r.set_map(cm, sm)
    eg, pg = g.associate(sc, ss, q, t); eo, po = o.associate(sc, ss, q, t)
    g.close(); o.close()
    assert eg.shape == eo.shape and pg.shape == po.shape and len(eg) > 40
    assert np.array_equal(eg[:, :3], eo[:, :3]) and np.array_equal(pg[:, :3], po[:, :3])
    if len(pg):
        assert np.abs(pg[:, 3:] - po[:, 3:]).max() < 1e-8


def test_interleaved_calls_and_map_updates(hip, setup):
    """align / associate / set_map interleaved on one context (buffers grow and are reused): the registration is unchanged."""
    g, o, sc, ss, R, t = setup
    q0 = mapreg.quat_from_R(R @ _rot(0.004, -0.003, 0.008)); t0 = t + np.array([0.05, -0.04, 0.03])
    q1, t1, s1 = g.align(hip.ctx, sc, ss, q0, t0)
    e, p = g.associate(sc, ss, q0, t0)                         # host-side tables (first use allocates the pinned read-back)
    q2, t2, s2 = g.align(hip.ctx, sc, ss, q0, t0)
    e2, p2 = g.associate(np.concatenate([sc, sc]), np.concatenate([ss, ss]), q0, t0)      # a larger scan: buffers regrow
    q3, t3, s3 = g.align(hip.ctx, sc, ss, q0, t0)
    assert len(e2) == 2 * len(e) and len(p2) == 2 * len(p)
    for (qa, ta, sa) in ((q2, t2, s2), (q3, t3, s3)):
        assert (sa.n_edge, sa.n_plane, sa.iterations) == (s1.n_edge, s1.n_plane, s1.iterations)        # (counts of the second round)
        assert np.abs(ta - t1).max() < 1e-10 and np.abs(qa - q1).max() < 1e-10


@pytest.fixture(scope="module")
def window_path(setup):
    """The same registration routed through the window solver (three launches per iteration): the cross-check of the one-launch pose solve."""
    g, o, sc, ss, R, t = setup
    w = mapreg.MapReg(lib.load_vilsolve(), "vmap_")
    assert w.lib.vmap_set_fused_max(w.ctx, 0) == 0
    w.set_map(*[a for a in (mapreg.make_map(seed=4, n_surf=12000, n_corner=2000))])
    yield w
    w.close()


@pytest.mark.parametrize("kw", [dict(), dict(max_iterations=12), dict(max_iterations=0), dict(jacobi_scaling=0), dict(lidar_loss=abi.LOSS_NONE),
                                dict(lidar_loss=abi.LOSS_CAUCHY, lidar_loss_scale=0.3), dict(initial_radius=1e-3, max_iterations=10), dict(precision=1),
                                dict(function_tolerance=1e-12, parameter_tolerance=1e-14, max_iterations=25)])
def test_pose_solve_matches_window_solver_and_oracle(hip, setup, window_path, kw):
    """One-launch pose solve == window solver == CPU oracle: same iteration count, same termination path, same pose, for every option it reads."""
    g, o, sc, ss, R, t = setup
    opts = abi.default_options(**({"max_iterations": 4} | kw))
    q0 = mapreg.quat_from_R(R @ _rot(0.004, -0.003, 0.008)); t0 = t + np.array([0.05, -0.04, 0.03])
    qg, tg, sg = g.align(hip.ctx, sc, ss, q0, t0, opts)
    qw, tw, sw = window_path.align(hip.ctx, sc, ss, q0, t0, opts)
    assert (sg.rounds, sg.n_edge, sg.n_plane, sg.iterations) == (sw.rounds, sw.n_edge, sw.n_plane, sw.iterations), kw
    tol = 1e-6 if kw.get("precision") else 1e-9               # fp32 factor arithmetic: the two paths differ by float rounding, amplified by the solve
    assert abs(sg.final_cost - sw.final_cost) <= tol * max(sw.final_cost, 1e-30) and abs(sg.initial_cost - sw.initial_cost) <= tol * max(sw.initial_cost, 1e-30)
    assert np.abs(tg - tw).max() < tol and np.abs(qg - qw).max() < tol
    if not kw.get("precision"):
        qo, to, so = o.align(None, sc, ss, q0, t0, opts)
        assert (sg.n_edge, sg.n_plane, sg.iterations) == (so.n_edge, so.n_plane, so.iterations), kw
        assert np.abs(tg - to).max() < 1e-8 and np.abs(qg - qo).max() < 1e-9


@pytest.mark.parametrize("nc,ns", [(0, 0), (0, 300), (60, 0), (3, 40), (400, 2500)])
def test_pose_solve_scan_shapes(hip, setup, window_path, nc, ns):
    """Empty, one-class and tiny scans (one workgroup) through the single-submission path."""
    g, o, sc, ss, R, t = setup
    q0 = mapreg.quat_from_R(R @ _rot(0.002, -0.001, 0.004)); t0 = t + np.array([0.02, -0.02, 0.01])
    qg, tg, sg = g.align(hip.ctx, sc[:nc], ss[:ns], q0, t0)
    qw, tw, sw = window_path.align(hip.ctx, sc[:nc], ss[:ns], q0, t0)
    assert (sg.rounds, sg.n_edge, sg.n_plane, sg.iterations) == (sw.rounds, sw.n_edge, sw.n_plane, sw.iterations)
    assert np.abs(tg - tw).max() < 1e-9 and np.abs(qg - qw).max() < 1e-10
    if nc + ns == 0:
        assert np.array_equal(tg, t0) and sg.iterations == 0


def test_pose_solve_many_in_a_row(hip, setup):
    """The workgroups of consecutive launches meet through the same block of device memory: 300 registrations, every one the same answer."""
    g, o, sc, ss, R, t = setup
    q0 = mapreg.quat_from_R(R @ _rot(0.004, -0.003, 0.008)); t0 = t + np.array([0.05, -0.04, 0.03])
    q1, t1, s1 = g.align(hip.ctx, sc, ss, q0, t0)
    for _ in range(300):
        q2, t2, s2 = g.align(hip.ctx, sc, ss, q0, t0)
        assert s2.iterations == s1.iterations and np.abs(t2 - t1).max() < 1e-12 and np.abs(q2 - q1).max() < 1e-12


def test_registration_concurrent_with_window_solver(hip, setup):
    """lidar_mapping and the estimator share the GPU: registrations (spinning cooperative workgroups) on one thread, window
    solves (helper workgroups spinning on their own launch) on another, different streams -- neither starves the other and
    both keep their answers."""
    import threading
    from mvil_fusion_amd import synth
    g, o, sc, ss, R, t = setup
    q0 = mapreg.quat_from_R(R @ _rot(0.004, -0.003, 0.008)); t0 = t + np.array([0.05, -0.04, 0.03])
    solver_a = lib.open_vilsolve()                                   # the registration's fallback solver handle (unused on this path)
    solver_b = lib.open_vilsolve()
    w = synth.make_config(2, L=300, n_plane=6000, n_edge=1500)
    opts = abi.default_options()
    solver_b.upload(w); ref = solver_b.solve_resident(opts)
    q1, t1, s1 = g.align(solver_a.ctx, sc, ss, q0, t0)
    out = {"reg": [], "sol": []}

    def reg():
        for _ in range(150):
            q2, t2, s2 = g.align(solver_a.ctx, sc, ss, q0, t0)
            out["reg"].append((s2.iterations, float(np.abs(t2 - t1).max())))

    def sol():
        for _ in range(150):
            solver_b.reset_state(); s = solver_b.solve_resident(opts)
            out["sol"].append((s.iterations, s.final_cost))

    th = [threading.Thread(target=reg), threading.Thread(target=sol)]
    for x in th: x.start()
    for x in th: x.join(timeout=120)
    assert not any(x.is_alive() for x in th), "a launch starved"
    assert len(out["reg"]) == 150 and all(it == s1.iterations and d < 1e-12 for it, d in out["reg"])
    assert len(out["sol"]) == 150 and all(it == ref.iterations and abs(c - ref.final_cost) <= 1e-9 * ref.final_cost for it, c in out["sol"])
    solver_a.close(); solver_b.close()

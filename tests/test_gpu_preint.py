"""GPU parity of the IMU pre-integration (include/vilpreint.h) against the CPU oracle, through the C-ABI."""
import numpy as np
import pytest

from mvil_fusion_amd import lib, preint

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,samples,seed", [(9, (20, 40), 0), (19, (5, 70), 1), (1, (1, 1), 2), (4, (16, 17), 3)])
def test_parity(oracle, n, samples, seed):
    s = preint.make_stream(n_intervals=n, samples=samples, seed=seed)
    g = preint.Preint(lib.load_vilsolve(), "vpre_"); o = preint.Preint(oracle.lib, "orc_vpre_")
    rg, jg = g.integrate(*s); ro, jo = o.integrate(*s)
    g.close(); o.close()
    assert np.abs(rg[:, :17] - ro[:, :17]).max() <= 1e-13                      # state chain: same operations, same order
    assert np.abs(jg - jo).max() <= 1e-12 * max(1.0, np.abs(jo).max())
    assert np.abs(rg[:, 62:] - ro[:, 62:]).max() <= 1e-12 * np.abs(ro[:, 62:]).max()
    assert np.abs(rg[:, 17:62] - ro[:, 17:62]).max() <= 1e-12 * max(1.0, np.abs(ro[:, 17:62]).max())


def test_empty_and_mixed_intervals(oracle):
    s = preint.make_stream(n_intervals=3, samples=(10, 10), seed=5)
    start = np.array([0, 10, 10, 30], np.int32)                          # middle interval without samples
    g = preint.Preint(lib.load_vilsolve(), "vpre_"); o = preint.Preint(oracle.lib, "orc_vpre_")
    rg, jg = g.integrate(start, *s[1:]); ro, jo = o.integrate(start, *s[1:])
    assert np.array_equal(rg[1], ro[1]) and np.array_equal(jg[1], np.eye(15))
    assert np.abs(rg - ro).max() <= 1e-12 * np.abs(ro).max()
    rec, jac = g.integrate(np.array([0], np.int32), s[1][:0], s[2][:0], s[3][:0], s[4][:0], s[5][:0], s[6][:0], s[7][:0])      # n = 0
    assert len(rec) == 0
    g.close(); o.close()


def test_records_drive_the_imu_factor(hip, oracle):
    """A device-produced record is a valid IMU factor constant: the IMU residuals / Jacobians evaluated with it equal those
    evaluated with the CPU-produced record of the same stream."""
    from mvil_fusion_amd import abi, synth
    s = preint.make_stream(n_intervals=9, samples=(20, 30), seed=7)
    g = preint.Preint(lib.load_vilsolve(), "vpre_"); o = preint.Preint(oracle.lib, "orc_vpre_")
    rg, _ = g.integrate(*s, want_jacobian=False); ro, _ = o.integrate(*s, want_jacobian=False)
    g.close(); o.close()
    wa, wb = synth.make_config(1), synth.make_config(1)
    n = min(len(rg), len(wa.imu_i))
    wa.imu_const[:n] = rg[:n]; wb.imu_const[:n] = ro[:n]
    ra, Ja = hip.eval_factors(wa, abi.FACTOR_IMU); rb, Jb = oracle.eval_factors(wb, abi.FACTOR_IMU)
    assert np.abs(ra - rb).max() <= 1e-6 * max(1.0, np.abs(rb).max()) and np.abs(Ja - Jb).max() <= 1e-6 * max(1.0, np.abs(Jb).max())


def test_long_intervals_and_degenerate_samples(oracle):
    """Intervals far longer than one batch (several hundred samples), zero-length time steps, large biases."""
    s = list(preint.make_stream(n_intervals=3, samples=(150, 400), seed=11))
    s[1] = s[1].copy(); s[1][::17] = 0.0                                  # dt = 0 samples
    s[6] = s[6] * 20.0; s[7] = s[7] * 20.0                                # biases of 1 m/s^2 and 0.1 rad/s
    g = preint.Preint(lib.load_vilsolve(), "vpre_"); o = preint.Preint(oracle.lib, "orc_vpre_")
    rg, jg = g.integrate(*s); ro, jo = o.integrate(*s)
    g.close(); o.close()
    assert np.abs(rg[:, :17] - ro[:, :17]).max() <= 1e-11 * max(1.0, np.abs(ro[:, :17]).max())
    assert np.abs(jg - jo).max() <= 1e-10 * np.abs(jo).max() and np.abs(rg[:, 62:] - ro[:, 62:]).max() <= 1e-10 * np.abs(ro[:, 62:]).max()

"""ctypes layer over include/vilpreint.h (IMU pre-integration) + a synthetic IMU stream.

`Preint(cdll, "vpre_")` drives csrc/libvilsolve.so (HIP; needs a GPU, no CPU fallback); `Preint(cdll, "orc_vpre_")` drives
oracle/liboracle.so -- tests / bench cpu_baseline leg only.
"""
import ctypes as C

import numpy as np

from . import synth

_dp, _ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
NOISE = np.array([synth.ACC_N, synth.GYR_N, synth.ACC_W, synth.GYR_W])


class PreintError(RuntimeError):
    pass


class Preint:
    def __init__(self, cdll, prefix="vpre_", device=0):
        self.lib, self.prefix = cdll, prefix
        self.ctx = C.c_void_p()
        f = getattr(cdll, prefix + "create"); f.restype = C.c_int
        st = f(C.c_int32(device), C.byref(self.ctx))
        if st != 0:
            self.ctx = None
            raise PreintError("%screate failed: status %d (no HIP device? there is no CPU fallback)" % (prefix, st))

    def close(self):
        if self.ctx is not None:
            f = getattr(self.lib, self.prefix + "destroy"); f.restype = None
            f(self.ctx); self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def bind(self, start, dt, acc, gyr, acc0, gyr0, ba, bg, noise=NOISE, want_jacobian=True):
        """Pre-builds the C arguments; returns (call, records, jacobians): call() runs the integration into the two arrays.
        bench.py times call() so that the Python marshalling is not part of the measurement."""
        start = np.ascontiguousarray(start, np.int32); n = len(start) - 1
        a = [np.ascontiguousarray(x, np.float64) for x in (dt, acc, gyr, acc0, gyr0, ba, bg, noise)]
        out = np.zeros((max(n, 1), 287)); jac = np.zeros((max(n, 1), 225)) if want_jacobian else None
        f = getattr(self.lib, self.prefix + "integrate"); f.restype = C.c_int
        args = [self.ctx, C.c_int32(n), start.ctypes.data_as(_ip)] + [x.ctypes.data_as(_dp) for x in a] + [out.ctypes.data_as(_dp), jac.ctypes.data_as(_dp) if want_jacobian else None]
        keep = (start, a)

        def call(_keep=keep):
            st = f(*args)
            if st != 0:
                raise PreintError("%sintegrate failed: status %d" % (self.prefix, st))
        return call, out[:n], (jac[:n].reshape(n, 15, 15) if want_jacobian else None)

    def integrate(self, start, dt, acc, gyr, acc0, gyr0, ba, bg, noise=NOISE, want_jacobian=True):
        """Returns (records n x 287, jacobians n x 15 x 15 or None)."""
        call, out, jac = self.bind(start, dt, acc, gyr, acc0, gyr0, ba, bg, noise, want_jacobian)
        call()
        return out, jac


def make_stream(n_intervals=9, samples=(20, 40), seed=0, rate=200.0):
    """A smooth synthetic IMU stream cut into intervals of varying length (one per consecutive frame pair of a window)."""
    rng = np.random.default_rng(seed)
    counts = rng.integers(samples[0], samples[1] + 1, n_intervals)
    start = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    ns = int(start[-1])
    tt = np.arange(ns + n_intervals + 1) / rate
    ph = rng.uniform(0, 2 * np.pi, (2, 3))
    acc_all = np.stack([0.8 * np.sin(1.3 * tt + ph[0, 0]), 0.6 * np.cos(0.9 * tt + ph[0, 1]), 9.8 + 0.4 * np.sin(2.1 * tt + ph[0, 2])], axis=1) + rng.normal(0, 0.05, (len(tt), 3))
    gyr_all = np.stack([0.3 * np.sin(0.7 * tt + ph[1, 0]), 0.25 * np.cos(1.1 * tt + ph[1, 1]), 0.4 * np.sin(0.5 * tt + ph[1, 2])], axis=1) + rng.normal(0, 0.01, (len(tt), 3))
    dt = np.full(ns, 1.0 / rate) * rng.uniform(0.9, 1.1, ns)
    acc = np.zeros((ns, 3)); gyr = np.zeros((ns, 3)); acc0 = np.zeros((n_intervals, 3)); gyr0 = np.zeros((n_intervals, 3))
    p = 0
    for k in range(n_intervals):
        acc0[k], gyr0[k] = acc_all[p], gyr_all[p]
        c = counts[k]
        acc[start[k]:start[k + 1]] = acc_all[p + 1:p + 1 + c]; gyr[start[k]:start[k + 1]] = gyr_all[p + 1:p + 1 + c]
        p += c                                                   # the last sample of an interval is the first measurement of the next
    ba = rng.normal(0, 0.05, (n_intervals, 3)); bg = rng.normal(0, 0.005, (n_intervals, 3))
    return start, dt, acc, gyr, acc0, gyr0, ba, bg

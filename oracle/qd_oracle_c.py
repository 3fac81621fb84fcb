"""ctypes binding of oracle/libqdoracle.so (the plain-C oracle).  TEST
INFRASTRUCTURE ONLY -- see the header of qd_oracle.c."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libqdoracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libqdoracle.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
        dp = ctypes.POINTER(ctypes.c_double); ip = ctypes.POINTER(ctypes.c_int32)
        _LIB.qdo_csd_channel_vc.restype = ctypes.c_int
        _LIB.qdo_csd_channel_vc.argtypes = [ctypes.c_int, ctypes.c_int, dp, dp, dp, dp, ctypes.c_double,
                                            ctypes.c_double, dp, dp, dp, ctypes.c_double, dp,
                                            ctypes.c_double, ctypes.c_int, ip, ip, dp, dp, dp,
                                            ctypes.c_int, ctypes.c_int, dp]
        _LIB.qdo_normalise.restype = ctypes.c_int
        _LIB.qdo_normalise.argtypes = [dp, ctypes.c_long, ctypes.POINTER(ctypes.c_float), dp]
        _LIB.qdo_env_images_vc.restype = ctypes.c_int
        _LIB.qdo_env_images_vc.argtypes = [ctypes.c_int, ctypes.c_int, dp, dp, dp, dp, ctypes.c_double,
                                           ctypes.c_double, dp, dp, dp, ctypes.c_double, dp,
                                           ctypes.c_double, dp, dp, dp, dp]
    return _LIB


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _vc(dev):
    """(alpha, beta) of the linear voltage-dependent capacitance model, or a NULL pointer."""
    vc = getattr(dev, "vc", None)
    return (None, None) if vc is None else _d(np.array(vc, dtype=np.float64))


def channel_gamma(dev, gate_v, ch):
    """Peak width of channel `ch`: constant, or utils/vary_peak_width.py with the current virtual
    plunger voltages (qarray_base_class.py:192-196)."""
    a = getattr(dev, "vpw_alpha", None)
    if a is None:
        return float(dev.gamma)
    v_avg = (abs(float(gate_v[ch])) + abs(float(gate_v[ch + 1]))) / 2
    return float(np.clip(dev.gamma - np.abs(a * v_avg), 0, 1))


def csd_channel(dev, vgm, origin, gate_v, sensor_v, barrier_v, window, ch, R,
                want_states=True, pix=None):
    """dev: qd_oracle.Device.  Returns dict(states, floors, occ, z, tc)."""
    N = dev.n_dot; P = R * R
    keep = [_d(dev.cdd_inv_full), _d(dev.cgd_full), _d(dev.Cbg), _d(dev.alpha), _d(vgm), _d(origin),
            _d(gate_v), _d(barrier_v)]
    states = np.zeros((P, 32, N), np.int32); floors = np.zeros((P, N), np.int32)
    occ = np.zeros((P, N)); z = np.zeros(P); tc = np.zeros((P, N - 1))
    ipt = ctypes.POINTER(ctypes.c_int32); dpt = ctypes.POINTER(ctypes.c_double)
    b, e = (0, -1) if pix is None else pix
    vc = _vc(dev)
    rc = lib().qdo_csd_channel_vc(N, R, keep[0][1], keep[1][1], keep[2][1], keep[3][1], float(dev.tc_base),
                                  channel_gamma(dev, gate_v, ch), keep[4][1], keep[5][1], keep[6][1], float(sensor_v),
                                  keep[7][1], float(window), int(ch),
                                  states.ctypes.data_as(ipt), floors.ctypes.data_as(ipt),
                                  occ.ctypes.data_as(dpt), z.ctypes.data_as(dpt), tc.ctypes.data_as(dpt),
                                  int(b), int(e), vc[1])
    if rc:
        raise RuntimeError(f"qdo_csd_channel rc={rc}")
    return dict(states=states, floors=floors, occ=occ, z=z, tc=tc)


def env_images(dev, vgm, origin, gate_v, sensor_v, barrier_v, window, R, want_occ=False):
    N = dev.n_dot; C = N - 1
    keep = [_d(dev.cdd_inv_full), _d(dev.cgd_full), _d(dev.Cbg), _d(dev.alpha), _d(vgm), _d(origin),
            _d(gate_v), _d(barrier_v)]
    z = np.zeros((C, R * R)); occ = np.zeros((C, R * R, N)) if want_occ else None
    dpt = ctypes.POINTER(ctypes.c_double)
    vc = _vc(dev)
    gam = _d(np.array([channel_gamma(dev, gate_v, ch) for ch in range(C)]))
    rc = lib().qdo_env_images_vc(N, R, keep[0][1], keep[1][1], keep[2][1], keep[3][1], float(dev.tc_base),
                                 float(dev.gamma), keep[4][1], keep[5][1], keep[6][1], float(sensor_v),
                                 keep[7][1], float(window), z.ctypes.data_as(dpt),
                                 occ.ctypes.data_as(dpt) if want_occ else None, vc[1], gam[1])
    if rc:
        raise RuntimeError(f"qdo_env_images rc={rc}")
    return (z, occ) if want_occ else z


def normalise(z):
    z = np.ascontiguousarray(z, dtype=np.float64)
    out = np.zeros(z.shape, np.float32); pl = np.zeros(2)
    rc = lib().qdo_normalise(z.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), z.size,
                             out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                             pl.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    if rc:
        raise RuntimeError(f"qdo_normalise rc={rc}")
    return out, pl

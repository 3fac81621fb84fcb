"""
ORACLE of the stochastic stages -- TEST INFRASTRUCTURE, like the rest of oracle/.

The reference draws these stages from numpy's GLOBAL generator (qarray WhiteNoise / TelegraphNoise /
LatchingModel, source absent; qarray_base_class.py:444-493), so sample-level parity with the reference is
impossible by construction.  What CAN be pinned is that the HIP kernels compute exactly the documented rules on
exactly the documented random stream.  This file restates, in numpy and independently of the kernels:

  * Philox4x32-10 (Salmon et al., SC'11) and the uniform / Box-Muller maps of csrc/qd_rng.h, keyed by
    (rng_seed folded to 32 bits, global env id) with counter (pixel, channel | purpose << 16, observation number);
  * the two-state telegraph chain along the raster (qd_k_telegraph);
  * the sensor stage with input noise on the sensor potential, LITERALLY as TunnelCoupledChargeSensed.py:342-380
    (eleven free energies, first differences, Lorentzians) -- not the kernel's closed form;
  * the radial image noise and the white-noise replacement, literally as qarray_base_class.py:444-493;
  * the latching walk as documented for qarray's LatchingModel.add_latching (ground_state.py:164):
    serial row-major walk, reset at each row, single-dot changes accepted with p_leads[dot], two-dot changes with
    p_inter[a][b], everything else always; a rejected pixel keeps the held occupations.  UNVERIFIED against qarray.

Parity status: "parity unpinned" against the reference (qarray source absent); pinned against this restatement.
"""
from __future__ import annotations

import numpy as np

import qd_oracle as O

RNG_WHITE, RNG_RADIAL, RNG_TELEGRAPH, RNG_LATCH = 1, 2, 3, 4
_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10.  Counter words broadcast against each other; returns four uint32 arrays."""
    c0, c1, c2, c3 = np.broadcast_arrays(*(np.asarray(c, dtype=np.uint32) for c in (c0, c1, c2, c3)))
    c0, c1, c2, c3 = c0.copy(), c1.copy(), c2.copy(), c3.copy()
    k0 = np.uint32(k0); k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = c0.astype(np.uint64) * _M0
            p1 = c2.astype(np.uint64) * _M1
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32); lo0 = p0.astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32); lo1 = p1.astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32((int(k0) + int(_W0)) & 0xFFFFFFFF); k1 = np.uint32((int(k1) + int(_W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def u01(a, b):
    """uniform in (0,1) from 53 random bits, never 0 (csrc/qd_rng.h::qd_u01)."""
    x = ((a.astype(np.uint64) << np.uint64(32)) | b.astype(np.uint64)) >> np.uint64(11)
    return (x.astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def normal2(v):
    u1 = u01(v[0], v[1]); u2 = u01(v[2], v[3])
    r = np.sqrt(-2.0 * np.log(u1)); t = 6.283185307179586476925 * u2
    return r * np.cos(t), r * np.sin(t)


class Stream:
    """The Philox key / counter convention of libqdsim for one env and one observation."""

    def __init__(self, rng_seed, global_env_id, obs_serial):
        s = int(rng_seed) & 0xFFFFFFFFFFFFFFFF
        self.k0 = (s ^ (s >> 32)) & 0xFFFFFFFF
        self.k1 = int(global_env_id) & 0xFFFFFFFF
        self.ser_lo = int(obs_serial) & 0xFFFFFFFF; self.ser_hi = (int(obs_serial) >> 32) & 0xFFFFFFFF

    def block(self, c0, ch, purpose):
        return philox4x32_10(c0, np.uint32(ch | (purpose << 16)), np.uint32(self.ser_lo), np.uint32(self.ser_hi), self.k0, self.k1)


def telegraph_bits(stream, ch, P, p01, p10):
    """State of the random telegraph process at every pixel of the raster (bool array of length P)."""
    v = stream.block(np.uint32(0xFFFFFFFF), ch, RNG_TELEGRAPH)
    pst = p01 / (p01 + p10) if p01 + p10 > 0 else 0.0
    state = bool(u01(v[0], v[1]) < pst)
    blocks = stream.block(np.arange((P + 1) // 2, dtype=np.uint32), ch, RNG_TELEGRAPH)
    ue = u01(blocks[0], blocks[1]); uo = u01(blocks[2], blocks[3])
    out = np.zeros(P, bool)
    for p in range(P):
        u = uo[p >> 1] if p & 1 else ue[p >> 1]
        if not state:
            state = bool(u < p01)
        elif u < p10:
            state = False
        out[p] = state
    return out


def latch_walk(stream, ch, occ, R, p_leads, p_inter):
    """occ (P,N) deterministic occupations -> latched occupations (documented LatchingModel behaviour)."""
    P, N = occ.shape
    v = stream.block(np.arange(P, dtype=np.uint32), ch, RNG_LATCH)
    u = u01(v[0], v[1])
    out = occ.copy()
    hold = None
    for p in range(P):
        nn = occ[p]
        accept = True
        if p % R != 0:
            differ = ~(np.abs(hold - nn) <= 1e-8 + 1e-5 * np.abs(nn))
            idx = np.nonzero(differ)[0]
            if len(idx) == 1:
                accept = bool(u[p] < p_leads[idx[0]])
            elif len(idx) == 2:
                accept = bool(u[p] < p_inter[idx[0], idx[1]])
        if accept:
            hold = nn.copy()
        else:
            out[p] = hold
    return out


def sensor_signal(dev, v_ext, n_open, gamma, eta):
    """TunnelCoupledChargeSensed.py:342-380 with input noise `eta` (P,) added to the sensor charge; output noise is
    zero (BaseNoiseModel default).  Literal: 2*n_peak+1 free energies, np.diff, Lorentzians."""
    N = dev.n_dot
    N_cont = v_ext @ dev.cgd_full.T
    N_sensor = np.round(N_cont[..., N:N + 1])
    F = []
    for k in range(-O.N_PEAK, O.N_PEAK + 1):
        q = np.concatenate([n_open, N_sensor + k + eta[:, None]], axis=-1)
        d = q - N_cont
        F.append(np.einsum('...i,ij,...j', d, dev.cdd_inv_full, d))
    dF = np.diff(np.stack(F), axis=0)
    return (1.0 / ((dF / gamma) ** 2 + 1.0)).sum(axis=0)


def radial_replaced(v1, v2, gt1, gt2, full):
    return full is not None and full > 0 and (abs(v1 - gt1) > full or abs(v2 - gt2) > full)


def observe_channel(dev, noise, stream, ch, R, vgm, origin, gate_v, sensor_v, barrier_v, window, gate_gt,
                    occ_det, flags, p_leads=None, p_inter=None, gamma=None):
    """One CSD channel with the stochastic stages `flags` ({"sensor","radial","latch"}) on.
    occ_det: (P,N) deterministic occupations (from the deterministic oracle).  noise: dict with white_amp, tel_p01,
    tel_p10, tel_amp, zero_radius, ramp_distance, full_noise_distance, max_amplitude.
    Returns (raw signal (P,), occupations used (P,N))."""
    P = R * R; N = dev.n_dot
    v1, v2 = gate_v[ch], gate_v[ch + 1]
    if "radial" in flags and radial_replaced(v1, v2, gate_gt[ch], gate_gt[ch + 1], noise["full_noise_distance"]):
        v = stream.block(np.arange(P, dtype=np.uint32), ch, RNG_RADIAL)
        return normal2(v)[0], occ_det                        # qarray_base_class.py:466-468: pure randn image
    occ = occ_det
    if "latch" in flags:
        occ = latch_walk(stream, ch, occ_det, R, p_leads, p_inter)
    vg = O.sweep_voltages(vgm, origin, gate_v, sensor_v, ch, -window, window, R)
    v_ext = np.concatenate([vg, np.broadcast_to(np.asarray(barrier_v, float), (P, N - 1))], axis=1)
    eta = np.zeros(P)
    if "sensor" in flags:
        v = stream.block(np.arange(P, dtype=np.uint32), ch, RNG_WHITE)
        eta = noise["white_amp"] * normal2(v)[0]
        eta = eta + noise["tel_amp"] * telegraph_bits(stream, ch, P, noise["tel_p01"], noise["tel_p10"])
    z = sensor_signal(dev, v_ext, occ, dev.gamma if gamma is None else gamma, eta)
    if "radial" in flags:
        alpha = noise["max_amplitude"] / noise["ramp_distance"]
        xs = np.linspace(v1 - window, v1 + window, R); ys = np.linspace(v2 - window, v2 + window, R)
        V1, V2 = np.meshgrid(xs, ys)
        dist = np.sqrt((V1 - gate_gt[ch]) ** 2 + (V2 - gate_gt[ch + 1]) ** 2).reshape(-1)
        amp = np.clip(alpha * (dist - noise["zero_radius"]), 0, noise["max_amplitude"])
        v = stream.block(np.arange(P, dtype=np.uint32), ch, RNG_RADIAL)
        z = z + normal2(v)[0] * amp
    return z, occ

"""CPU tier: the numpy restatement of the random stream (oracle/qd_noise_oracle.py) against the Random123
known-answer vectors and against the product's own Philox / uniform / normal code compiled for the host."""
import ctypes

import numpy as np

import helpers as H
import qd_noise_oracle as NO


def test_numpy_philox_known_answers():
    f = 0xFFFFFFFF
    kat = [((0, 0, 0, 0, 0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((f, f, f, f, f, f), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, 0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for args, want in kat:
        got = NO.philox4x32_10(*args)
        assert tuple(int(x) for x in got) == want


def test_numpy_stream_equals_product_code():
    """Same blocks, uniforms and normals as csrc/qd_rng.h (host build), for a range of counters."""
    h = H.hosttest()
    out = (ctypes.c_uint32 * 4)()
    s = NO.Stream(rng_seed=0x123456789ABCDEF, global_env_id=7, obs_serial=3)
    for p in (0, 1, 2, 1000, 4095):
        v = s.block(np.uint32(p), 2, NO.RNG_WHITE)
        h.qdh_philox(p, 2 | (NO.RNG_WHITE << 16), s.ser_lo, s.ser_hi, s.k0, s.k1, out)
        assert [int(x) for x in v] == list(out)
    a = np.array([0, 1, 0xFFFFFFFF, 12345], np.uint32); b = np.array([0, 0xFFFFFFFF, 0xFFFFFFFF, 678], np.uint32)
    u = NO.u01(a, b)
    assert np.all((u > 0) & (u <= 1)) and u[0] == 0.5 / 2 ** 53        # never 0 (log), 1.0 only by rounding
    z0, z1 = NO.normal2(tuple(np.array([x], np.uint32) for x in (1, 2, 3, 4)))
    assert np.isfinite(z0).all() and np.isfinite(z1).all()


def test_telegraph_chain_is_a_two_state_markov_chain():
    s = NO.Stream(5, 0, 1)
    bits = NO.telegraph_bits(s, 1, 20000, 0.02, 0.05)
    up = np.mean(bits); flips = np.mean(bits[1:] != bits[:-1])
    assert abs(up - 0.02 / 0.07) < 0.05 and abs(flips - 2 * 0.02 * 0.05 / 0.07) < 0.01

"""The per-pixel __host__ __device__ code of the HIP kernels (csrc/qd_pixel.h),
compiled for the CPU, against the oracle: candidate charge states and floors
bit-exact, couplings and sensor signal to round-off.  No GPU needed."""
import numpy as np
import pytest

import qd_oracle_c as OC
import helpers as H
from qadapt_hip.layout import layout, LAYOUT_FIELDS


@pytest.mark.parametrize("N", [2, 3, 4, 5, 6, 7, 8])
def test_layout_python_mirror_matches_c(N):
    c = H.host_layout(N)
    L = layout(N)
    for f in LAYOUT_FIELDS:
        assert c[f] == getattr(L, f), f


@pytest.mark.parametrize("N,R,mode", [(2, 16, "near"), (2, 16, "far"), (3, 12, "mid"), (4, 16, "near"),
                                       (4, 12, "far"), (5, 8, "mid"), (6, 6, "near"), (7, 4, "mid"),
                                       (8, 4, "near"), (8, 3, "far")])
def test_candidates_bit_exact_vs_oracle(N, R, mode):
    eb = H.sample_blocks(N, [4000 + N, 4100 + N])
    rng = np.random.default_rng(N * 31 + len(mode))
    for k in range(2):
        par = eb.params[k]; st = H.place(N, eb.state[k], mode, rng)
        dev = H.dev_view(N, par); sv = H.state_view(N, st)
        for ch in sorted({0, N - 2}):
            got = H.host_front(N, par, st, ch, R)
            ref = OC.csd_channel(dev, sv.vgm, dev.origin, sv.gate_v, sv.sensor_gt, sv.barrier_v,
                                 dev.window, ch, R)
            assert np.array_equal(got["floors"], ref["floors"])
            assert np.array_equal(got["states"], ref["states"])
            assert np.allclose(got["tc"], ref["tc"], rtol=1e-13)
            # sensor stage (closed-form differences) on the oracle's occupations
            for p in range(0, R * R, max(1, R * R // 7)):
                z = H.host_sensor(N, par, got["vpp"][p], ref["occ"][p])
                assert np.isclose(z, ref["z"][p], rtol=1e-9, atol=1e-12)


def test_search_prunes(capsys):
    # the exact search must visit far fewer leaves than the 4^8 brute force
    N, R = 8, 6
    eb = H.sample_blocks(N, [77])
    st = H.place(N, eb.state[0], "near", np.random.default_rng(1))
    got = H.host_front(N, eb.params[0], st, 3, R)
    nodes, leaves, inserts, shifts = (got["stats"] / (R * R)).tolist()
    print(f"N=8 search per pixel: nodes {nodes:.0f} leaves {leaves:.0f} inserts {inserts:.0f}")
    assert leaves < 4 ** 8 / 20


def test_philox_known_answers_and_normals():
    """Philox4x32-10 known-answer vectors (Random123 kat_vectors) and the Box-Muller transform
    used by the stochastic stages (csrc/qd_rng.h), on the CPU build of the device code."""
    import ctypes
    h = H.hosttest()
    out = (ctypes.c_uint32 * 4)()
    h.qdh_philox(0, 0, 0, 0, 0, 0, out)
    assert [hex(v) for v in out] == ['0x6627e8d5', '0xe169c58d', '0xbc57ac4c', '0x9b00dbd8']
    f = 0xFFFFFFFF
    h.qdh_philox(f, f, f, f, f, f, out)
    assert [hex(v) for v in out] == ['0x408f276d', '0x41c83b0e', '0xa20bc7c6', '0x6d5451fd']
    h.qdh_philox(0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, 0xa4093822, 0x299f31d0, out)
    assert [hex(v) for v in out] == ['0xd16cfe09', '0x94fdcceb', '0x5001e420', '0x24126ea1']
    n = 200000
    z = np.zeros(n)
    h.qdh_normals(123, 456, n, z.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01
    assert abs(np.mean(z ** 3)) < 0.03 and abs(np.mean(z ** 4) - 3) < 0.08
    assert abs(np.corrcoef(z[0::2], z[1::2])[0, 1]) < 0.01

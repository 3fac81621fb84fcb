"""The plain-C oracle (fast, OpenMP; CPU baseline of bench.py) must agree with
the literal NumPy oracle: integers exactly, floats to round-off."""
import numpy as np
import pytest

import qd_oracle as O
import qd_oracle_c as OC


def _scene(N, seed, mode):
    rng = np.random.default_rng(seed)
    s = O.sample_episode(rng, N)
    dev = O.device_from_sample(s)
    vgm = O.identity_vgm(N) + rng.normal(0, 0.05, (N + 1, N + 1))
    origin = np.zeros(N + 1)
    pgt, bgt, sgt = O.ground_truth(dev, vgm, origin)
    span = {"near": (3, 3), "mid": (10, 6), "far": (40, 12)}[mode]
    gv = pgt.astype(float) + rng.uniform(-span[0], span[0], N)
    bv = bgt.astype(float) + rng.uniform(-span[1], span[1], N - 1)
    return dev, vgm, origin, gv, bv, sgt, s["window_delta"]


@pytest.mark.parametrize("N,R,mode", [(2, 12, "near"), (2, 12, "far"), (3, 10, "mid"),
                                       (4, 10, "near"), (4, 10, "mid"), (5, 6, "mid"), (6, 4, "near")])
def test_c_matches_numpy_channel(N, R, mode):
    dev, vgm, origin, gv, bv, sgt, w = _scene(N, 7 * N + len(mode), mode)
    for ch in sorted({0, N - 2}):
        c = OC.csd_channel(dev, vgm, origin, gv, sgt, bv, w, ch, R)
        vg = O.sweep_voltages(vgm, origin, gv, sgt, ch, -w, w, R)
        vb = np.broadcast_to(bv, (R * R, N - 1))
        n, st, F, tc = O.ground_state_open(dev, vg, vb, return_states=True)
        z, _ = O.charge_sensor_open(dev, vg, vb, n_open=n)
        assert np.array_equal(c["states"], st)              # integer charge states: exact
        assert np.allclose(c["tc"], tc, rtol=1e-12)
        assert np.allclose(c["occ"], n, atol=1e-7)
        assert np.allclose(c["z"], z[:, 0], rtol=1e-6, atol=1e-8)


def test_c_8dot_small():
    N, R = 8, 3
    dev, vgm, origin, gv, bv, sgt, w = _scene(N, 99, "near")
    c = OC.csd_channel(dev, vgm, origin, gv, sgt, bv, w, 3, R)
    vg = O.sweep_voltages(vgm, origin, gv, sgt, 3, -w, w, R)
    vb = np.broadcast_to(bv, (R * R, N - 1))
    n, st, F, tc = O.ground_state_open(dev, vg, vb, return_states=True)
    assert np.array_equal(c["states"], st)
    assert np.allclose(c["occ"], n, atol=1e-7)


def test_c_normalise_matches_numpy():
    rng = np.random.default_rng(3)
    for shape in [(7, 64 * 64), (1, 32 * 32), (3, 100), (5, 144), (1, 577), (3, 4115), (1, 10)]:
        z = rng.normal(size=shape) ** 3
        out, pl = OC.normalise(z)
        assert pl[0] == np.percentile(z, 0.5) and pl[1] == np.percentile(z, 99.5)
        assert np.array_equal(out, O.normalise_image(z))
    out, _ = OC.normalise(np.ones((2, 50)))
    assert np.all(out == 0)


def test_c_env_images_layout():
    N, R = 3, 6
    dev, vgm, origin, gv, bv, sgt, w = _scene(N, 5, "near")
    z = OC.env_images(dev, vgm, origin, gv, sgt, bv, w, R)
    img = O.get_obs_images(dev, vgm, origin, gv, bv, sgt, w, R)       # (R,R,C)
    assert np.allclose(z.reshape(N - 1, R, R).transpose(1, 2, 0), img, rtol=1e-6, atol=1e-8)

"""Pin the oracle against fixtures produced by the reference's own importable
modules (tests/golden/make_golden.py): Kalman traces (a19) and sweep grids (a5)."""
import os

import numpy as np

import qd_oracle as O


def test_kalman_traces_match_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "kalman_traces.npz"))
    for c in range(int(g["n_cases"])):
        n = int(g[f"c{c}_n_dots"])
        k = O.KalmanOracle(n)
        values = g[f"c{c}_values"]; log_vars = g[f"c{c}_log_vars"]
        for t in range(values.shape[0]):
            k.update_from_cnn(values[t], log_vars[t])
            # same float64 operations in the same order -> bit-exact
            assert np.array_equal(k.means, g[f"c{c}_means"][t])
            assert np.array_equal(k.vars, g[f"c{c}_variances"][t])
            assert np.array_equal(k.full_matrix(), g[f"c{c}_full"][t])


def test_kalman_readme_example():
    # SURVEY §8c captured example
    k = O.KalmanOracle(4)
    k.update_from_scan(1, [(-0.05, -4.0), (0.02, -3.5), (0.01, -1.0)])
    m = k.full_matrix()
    assert abs(m[1, 2] - 0.25176684) < 1e-8
    assert abs(m[1, 3] - 0.1688609) < 1e-7
    assert m[0, 2] == 0.15


def test_sweep_grids_match_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "sweep_grids.npz"))
    for c in range(int(g["n_cases"])):
        R = int(g[f"c{c}_R"]); ch = int(g[f"c{c}_ch"]); w = float(g[f"c{c}_window"])
        vg = O.sweep_voltages(g[f"c{c}_vgm"], g[f"c{c}_origin"], g[f"c{c}_gate_voltages"],
                              float(g[f"c{c}_sensor"]), ch, -w, w, R)
        assert np.array_equal(vg, g[f"c{c}_vg_flat"])


def test_direct_and_nn_mode_updaters_match_reference(golden_dir):
    """update_method "direct" (DirectUpdater.py) and the legacy 2-output nearest_neighbour mode of both
    updaters, against traces of the reference classes themselves."""
    g = np.load(os.path.join(golden_dir, "updater_variants.npz"))
    for c in range(int(g["n_cases"])):
        n = int(g[f"c{c}_n_dots"]); kind = str(g[f"c{c}_kind"])
        values = g[f"c{c}_values"]; log_vars = g[f"c{c}_log_vars"]
        cls = O.DirectOracle if kind == "direct" else O.KalmanOracle
        k = cls(n, include_nnn=values.shape[-1] == 3)
        for t in range(values.shape[0]):
            k.update_from_cnn(values[t], log_vars[t])
            assert np.array_equal(k.means, g[f"c{c}_means"][t]), (c, t)
            assert np.array_equal(k.vars, g[f"c{c}_variances"][t]), (c, t)
            assert np.array_equal(k.full_matrix(), g[f"c{c}_full"][t]), (c, t)

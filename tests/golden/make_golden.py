"""
Generate the golden fixtures under tests/golden/ from the pieces of the
reference that ARE importable in the build container (numpy-only modules).

Run ONCE, in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

It loads these reference source files by path and executes them unmodified:
  * src/qadapt/capacitance_model/KalmanUpdater.py        
  * src/qadapt/capacitance_model/DirectUpdater.py        (update_method "direct")
  * src/qarray_latched/DotArrays/GateVoltageComposer.py  
  * src/qadapt/environment/utils/vary_peak_width.py      (variable peak width)
and stores only INPUTS and OUTPUTS (arrays) as .npz -- no reference source is
copied.  Everything else on the hot path imports jax/qarray and cannot run
here; those rows are pinned by analytic known-answer tests.
"""
import importlib.util
import os

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def kalman_traces():
    K = _load(f"{REF}/src/qadapt/capacitance_model/KalmanUpdater.py", "ref_kalman")
    out = {}
    for case, (n_dots, steps, seed) in enumerate([(2, 6, 1), (4, 12, 2), (6, 10, 3), (8, 25, 4)]):
        rng = np.random.default_rng(seed)
        # env.py:779-787 constructor arguments
        k = K.KalmanCapacitanceUpdater(n_dots=n_dots, prior_mean=0.3, prior_variance=0.5,
                                       variance_threshold=0.05, process_noise=0.0,
                                       include_nnn=True, prior_mean_nnn=0.15)
        C = n_dots - 1
        # CNN-like outputs; log-vars straddle the 0.05 gate (ln 0.05 = -2.996)
        # and the [-6, 2] clamp; a few large deltas hit the [-1, 1] mean clamp.
        values = rng.normal(0.0, 0.1, size=(steps, C, 3))
        values[rng.random(values.shape) < 0.05] *= 40.0
        log_vars = rng.uniform(-8.0, -1.5, size=(steps, C, 3))
        means = np.zeros((steps, n_dots, n_dots)); varis = np.zeros_like(means)
        full = np.zeros_like(means)
        for t in range(steps):
            for i in range(C):
                # env.py:610-618: predictions are negated by the caller
                outs = [(-float(values[t, i, j]), float(log_vars[t, i, j])) for j in range(3)]
                k.update_from_scan(left_dot=i, ml_outputs=outs)
            means[t] = k.means; varis[t] = k.variances; full[t] = k.get_full_matrix()
        out[f"c{case}_n_dots"] = np.array(n_dots)
        out[f"c{case}_values"] = values; out[f"c{case}_log_vars"] = log_vars
        out[f"c{case}_means"] = means; out[f"c{case}_variances"] = varis
        out[f"c{case}_full"] = full
        out[f"c{case}_accepted"] = np.array(k.total_accepted)
        out[f"c{case}_rejected"] = np.array(k.total_rejected)
    out["n_cases"] = np.array(4)
    np.savez_compressed(os.path.join(HERE, "kalman_traces.npz"), **out)


def updater_variant_traces():
    """DirectUpdater.py (update_method "direct") in 3-output mode and both updaters in the legacy
    2-output nearest_neighbour mode (include_nnn=False, env.py:784) -- the reference classes themselves."""
    K = _load(f"{REF}/src/qadapt/capacitance_model/KalmanUpdater.py", "ref_kalman")
    D = _load(f"{REF}/src/qadapt/capacitance_model/DirectUpdater.py", "ref_direct")
    out = {}
    cases = [("direct", D.DirectCapacitanceUpdater, 4, 10, 3, 11), ("direct", D.DirectCapacitanceUpdater, 8, 12, 3, 12),
             ("direct", D.DirectCapacitanceUpdater, 6, 8, 2, 13), ("kalman", K.KalmanCapacitanceUpdater, 4, 10, 2, 14),
             ("kalman", K.KalmanCapacitanceUpdater, 8, 9, 2, 15)]
    for case, (name, cls, n_dots, steps, n_out, seed) in enumerate(cases):
        rng = np.random.default_rng(seed)
        k = cls(n_dots=n_dots, prior_mean=0.3, prior_variance=0.5, variance_threshold=0.05, process_noise=0.0,
                include_nnn=(n_out == 3), prior_mean_nnn=0.15)
        C = n_dots - 1
        values = rng.normal(0.0, 0.1, size=(steps, C, n_out))
        values[rng.random(values.shape) < 0.05] *= 40.0
        log_vars = rng.uniform(-8.0, -1.5, size=(steps, C, n_out))
        means = np.zeros((steps, n_dots, n_dots)); varis = np.zeros_like(means); full = np.zeros_like(means)
        for t in range(steps):
            for i in range(C):
                outs = [(-float(values[t, i, j]), float(log_vars[t, i, j])) for j in range(n_out)]
                k.update_from_scan(left_dot=i, ml_outputs=outs)
            means[t] = k.means; varis[t] = k.variances; full[t] = k.get_full_matrix()
        out[f"c{case}_kind"] = np.array(name); out[f"c{case}_n_dots"] = np.array(n_dots)
        out[f"c{case}_values"] = values; out[f"c{case}_log_vars"] = log_vars
        out[f"c{case}_means"] = means; out[f"c{case}_variances"] = varis; out[f"c{case}_full"] = full
    out["n_cases"] = np.array(len(cases))
    np.savez_compressed(os.path.join(HERE, "updater_variants.npz"), **out)


def sweep_grids():
    G = _load(f"{REF}/src/qarray_latched/DotArrays/GateVoltageComposer.py", "ref_gvc")
    out = {}
    cases = [(2, 5, 0, 10), (4, 8, 1, 11), (4, 8, 2, 12), (8, 6, 0, 13), (8, 6, 6, 14)]
    for case, (n_dot, R, ch, seed) in enumerate(cases):
        rng = np.random.default_rng(seed)
        Gt = n_dot + 1
        vgm = -np.eye(Gt) + rng.normal(0, 0.2, size=(Gt, Gt))
        origin = rng.normal(0, 1.0, size=Gt)
        gv = rng.uniform(-30, 30, size=n_dot)
        sensor = float(rng.uniform(-1, 1))
        w = float(rng.uniform(1.5, 2.0))
        comp = G.GateVoltageComposer(n_gate=Gt, n_dot=n_dot, n_sensor=1)
        comp.virtual_gate_matrix = vgm
        comp.virtual_gate_origin = origin
        # qarray_base_class.py:114-154: exactly this call
        gate_voltages = np.concatenate([gv, [sensor]])
        v1 = gv[ch]; v2 = gv[ch + 1]
        vg = comp.do2d(f"vP{ch + 1}", v1 - w, v1 + w, R, f"vP{ch + 2}", v2 - w, v2 + w, R,
                       gate_voltages, True)
        out[f"c{case}_n_dot"] = np.array(n_dot); out[f"c{case}_R"] = np.array(R)
        out[f"c{case}_ch"] = np.array(ch)
        out[f"c{case}_vgm"] = vgm; out[f"c{case}_origin"] = origin
        out[f"c{case}_gate_voltages"] = gv; out[f"c{case}_sensor"] = np.array(sensor)
        out[f"c{case}_window"] = np.array(w)
        out[f"c{case}_vg_flat"] = vg.reshape(-1, vg.shape[-1])
    out["n_cases"] = np.array(len(cases))
    np.savez_compressed(os.path.join(HERE, "sweep_grids.npz"), **out)


def peak_widths():
    V = _load(f"{REF}/src/qadapt/environment/utils/vary_peak_width.py", "ref_vpw")
    rng = np.random.default_rng(21)
    n = 64
    w0 = rng.uniform(0.0, 0.4, n); alpha = rng.uniform(0.0001, 0.0008, n)
    alpha[:4] = [0.0, 0.01, -0.002, 0.05]                 # zero, the ctor default, negative, saturating
    vx = rng.uniform(-120, 120, n); vy = rng.uniform(-120, 120, n)
    w0[4] = 1.7                                           # clipped at the upper bound
    out = np.array([V.VaryPeakWidth(peak_width_0=w0[i], alpha=alpha[i]).linearly_vary_peak_width(vx[i], vy[i])
                    for i in range(n)], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "peak_widths.npz"), peak_width_0=w0, alpha=alpha, v_x=vx, v_y=vy, width=out)


if __name__ == "__main__":
    if "--only-peak-widths" in __import__("sys").argv:
        peak_widths()
        raise SystemExit(0)
    if "--only-updater-variants" in __import__("sys").argv:
        updater_variant_traces()
        raise SystemExit(0)
    kalman_traces()
    updater_variant_traces()
    sweep_grids()
    peak_widths()
    print("golden fixtures written to", HERE)

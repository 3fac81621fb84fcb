"""
Reset-time device sampling and parameter-block construction (host side, per
episode).  Vectorised over the envs being reset.

Mirrors, for the barrier model only (the reference's env supports nothing else,
src/qadapt/environment/env.py:61-62):
  * QarrayBaseClass._gen_random_qarray_params_with_barriers and its helpers
    (src/qadapt/environment/qarray_base_class.py:254-390, 495-555, 611-700)
  * TunnelCoupledChargeSensed.update_capacitance_matrices / Maxwell conversion
    (src/qarray_latched/DotArrays/TunnelCoupledChargeSensed.py:94-143,
     src/qarray_latched/DotArrays/_helper_functions.py:60-164)
  * optimal_Vg and calculate_ground_truth
    (TunnelCoupledChargeSensed.py:445-471, qarray_base_class.py:1233-1286)
  * QuantumDeviceEnv.reset's window / offset / range / start draws
    (src/qadapt/environment/env.py:160-206, 808-858)

Randomness: the reference draws from unseeded generators, so only the priors and
construction rules are contractual.  Here every env owns a numpy Generator
(PCG64(seed + env_id)); one episode consumes a fixed-length vector of uniforms
in the reference's draw order (see `DrawPlan`), which is what the oracle's
literal scalar sampler reproduces in tests/test_device_model.py.
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import numpy as np
import yaml

from .layout import layout

_CFG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs")


def load_yaml(path, default_name):
    if path is None:
        path = os.path.join(_CFG_DIR, default_name)
    elif not os.path.isabs(path) and not os.path.exists(path):
        path = os.path.join(_CFG_DIR, path)
    if not os.path.exists(path):
        raise FileNotFoundError(f"Config file not found: {path}")
    with open(path) as fh:
        return yaml.safe_load(fh)


def _mm(d):
    return float(d["min"]), float(d["max"])


def _band(tbl, dist):
    return _mm(tbl[1] if dist == 1 else tbl[2] if dist == 2 else tbl["3_plus"])


@dataclass
class DrawPlan:
    """The uniform draws of one episode, in order: lo/hi per draw and the named
    slices that say where each run of draws goes."""
    lo: np.ndarray
    hi: np.ndarray
    slices: dict
    index: dict           # name -> integer index arrays for scattering into matrices


def make_draw_plan(N, qcfg, ecfg) -> DrawPlan:
    m = qcfg["simulator"]["model"]
    sim = ecfg["simulator"]
    nb = N - 1
    lo, hi, slices, index = [], [], {}, {}

    def run(name, bounds):
        s = len(lo)
        for b in bounds:
            lo.append(b[0]); hi.append(b[1])
        slices[name] = slice(s, len(lo))

    run("window_delta", [_mm(sim["window_delta_range"])])
    # Cdd: upper triangle incl. diagonal, the diagonal draws nothing (:254-268)
    ij = [(i, j) for i in range(N) for j in range(i + 1, N)]
    run("Cdd", [_band(m["Cdd"]["distance_coupling"], j - i) for i, j in ij])
    index["sym_ut"] = (np.array([p[0] for p in ij], int), np.array([p[1] for p in ij], int))
    # Cgd: all N*N plunger entries row-major (:270-298)
    run("Cgd", [(_mm(m["Cgd"]["primary_coupling"]) if i == j else
                 _band(m["Cgd"]["cross_coupling"], abs(i - j))) for i in range(N) for j in range(N)])
    run("Cds", [_mm(m["Cds"]["dots"])] * N)
    run("Cgs", [_mm(m["Cgs"]["plunger_gates"])] * N + [_mm(m["Cgs"]["sensor_gate"])])
    run("Cbd", [_band(m["Cbd"]["distance_coupling"], max(1, int(abs(i - (j + 0.5)))))
                for i in range(N) for j in range(nb)])
    run("Cbg", [_band(m["Cbg"]["distance_coupling"],
                      max(1, int(abs((i + 0.5) - j))) if j < N else 2)
                for i in range(nb) for j in range(N + 1)])
    run("Cbs", [_mm(m["Cbs"]["coupling"])] * nb)
    bij = [(i, j) for i in range(nb) for j in range(i + 1, nb)]
    run("Cbb", [_band(m["Cbb"]["distance_coupling"], j - i) for i, j in bij])
    index["bb_ut"] = (np.array([p[0] for p in bij], int), np.array([p[1] for p in bij], int))
    run("white_noise_amplitude", [_mm(m["white_noise_amplitude"])])
    t = m["telegraph_noise_parameters"]
    run("telegraph", [_mm(t["p01"]), _mm(t["p10_factor"]), _mm(t["amplitude"])])
    rn = sim.get("radial_noise") or {"enabled": False}
    if rn.get("enabled"):
        def mm_or_const(v):
            return _mm(v) if isinstance(v, dict) else (float(v), float(v))
        bounds = []
        if isinstance(rn["lower"], dict):
            bounds.append(_mm(rn["lower"]))
        if isinstance(rn["ramp_range"], dict):
            bounds.append(_mm(rn["ramp_range"]))
        if isinstance(rn.get("total_noise_range"), dict):
            bounds.append(_mm(rn["total_noise_range"]))
        run("radial", bounds)
    else:
        run("radial", [])
    lt = m["latching_model_parameters"]
    run("p_inter", [_mm(lt["p_inter"])] * len(ij))
    run("p_leads", [_mm(lt["p_leads"])] * N)
    run("tc_base", [_mm(m["barrier_model"]["tc_base"])])
    run("alpha", [_mm(m["barrier_model"]["alpha_per_barrier"])] * nb)
    run("vcap", [_mm(m["voltage_capacitance_model"]["alpha"]), _mm(m["voltage_capacitance_model"]["beta"])])
    run("vpw_alpha", [_mm(m["variable_peak_width_model"]["alpha"])])
    run("T", [_mm(m["T"])])
    run("coulomb_peak_width", [_mm(m["coulomb_peak_width"])])
    run("tc", [_mm(m["tc"])])
    run("offset", [_mm(sim["constant_voltage_offset"])] * N)
    run("u_plunger_range", [_mm(sim["full_plunger_range_width"])])
    run("u_plunger_center", [(0.0, 1.0)] * N)
    run("u_barrier_range", [_mm(sim["full_barrier_range_width"])])
    run("u_barrier_center", [(0.0, 1.0)] * nb)
    run("u_start_plunger", [(0.0, 1.0)] * N)
    run("u_start_barrier", [(0.0, 1.0)] * nb)
    return DrawPlan(np.array(lo), np.array(hi), slices, index)


class EpisodeBatch:
    """Everything the device needs for a batch of freshly reset envs."""

    def __init__(self, n, N):
        L = layout(N)
        self.params = np.zeros((n, L.size))
        self.state = np.zeros((n, L.s_size))
        self.extras = {}


class DeviceSampler:
    def __init__(self, N, qcfg, ecfg, vary_peak_width=False, peak_width_alpha=0.01, perfect_vgm=False):
        self.N = N
        self.perfect_vgm = bool(perfect_vgm)          # update_method "perfect" (env.py:181-182)
        self.qcfg = qcfg
        self.ecfg = ecfg
        self.plan = make_draw_plan(N, qcfg, ecfg)
        self.L = layout(N)
        meas = qcfg["simulator"]["measurement"]
        self.optimal_tc = float(meas["tc"])
        c = meas["optimal_VG_center"]
        self.n_star = np.array([float(c["dots"])] * N + [float(c["sensor"])])
        self.cdd_diag = float(qcfg["simulator"]["model"]["Cdd"]["diagonal"])
        self.cbb_diag = float(qcfg["simulator"]["model"]["Cbb"]["diagonal"])
        if qcfg["simulator"]["model"].get("charge_carrier_type", "electrons") != "electrons":
            raise NotImplementedError("only charge_carrier_type 'electrons' (the reference default) is built")
        # f4: voltage-dependent capacitances (qarray_base_class.py:840-854) and variable peak width (:856-863)
        self.vc_type = qcfg["simulator"]["voltage_capacitance_model"]["type"]
        if self.vc_type not in (None, "linear"):
            raise ValueError(f"Capacitance model type '{self.vc_type}' does not exist")
        self.vary_peak_width = bool(vary_peak_width)
        self.peak_width_alpha = float(peak_width_alpha)

    # -- raw draws -> named arrays ------------------------------------------------
    def draws_from_uniform(self, u):
        """u: (n, n_draws) standard uniforms -> dict of named arrays (n, ...)."""
        p = self.plan
        v = p.lo + (p.hi - p.lo) * u
        return {k: v[:, s] for k, s in p.slices.items()}

    def assemble(self, d):
        """named draws -> capacitance matrices etc. (all batched on axis 0)."""
        N = self.N; nb = N - 1; n = d["Cgd"].shape[0]
        iu, ju = self.plan.index["sym_ut"]
        Cdd = np.zeros((n, N, N))
        Cdd[:, np.arange(N), np.arange(N)] = self.cdd_diag
        Cdd[:, iu, ju] = d["Cdd"]; Cdd[:, ju, iu] = d["Cdd"]
        Cgd = np.zeros((n, N, N + 1))
        raw = d["Cgd"].reshape(n, N, N)
        sym = (raw + raw.transpose(0, 2, 1)) / 2
        dg = np.arange(N)
        sym[:, dg, dg] = raw[:, dg, dg]
        Cgd[:, :, :N] = sym
        out = dict(Cdd=Cdd, Cgd=Cgd, Cds=d["Cds"].reshape(n, 1, N), Cgs=d["Cgs"].reshape(n, 1, N + 1),
                   Cbd=d["Cbd"].reshape(n, N, nb), Cbg=d["Cbg"].reshape(n, nb, N + 1),
                   Cbs=d["Cbs"].reshape(n, 1, nb))
        bi, bj = self.plan.index["bb_ut"]
        Cbb = np.zeros((n, nb, nb))
        Cbb[:, np.arange(nb), np.arange(nb)] = self.cbb_diag
        if len(bi):
            Cbb[:, bi, bj] = d["Cbb"]; Cbb[:, bj, bi] = d["Cbb"]
        out["Cbb"] = Cbb
        p_inter = np.zeros((n, N, N))
        p_inter[:, iu, ju] = d["p_inter"]; p_inter[:, ju, iu] = d["p_inter"]
        out.update(p_inter=p_inter, p_leads=d["p_leads"], tc_base=d["tc_base"][:, 0], alpha=d["alpha"],
                   coulomb_peak_width=d["coulomb_peak_width"][:, 0], window_delta=d["window_delta"][:, 0],
                   white_noise_amplitude=d["white_noise_amplitude"][:, 0],
                   telegraph=dict(p01=d["telegraph"][:, 0], p10=d["telegraph"][:, 1] * d["telegraph"][:, 0],
                                  amplitude=d["telegraph"][:, 2]),
                   offset=d["offset"], radial=d["radial"],
                   vc_alpha=d["vcap"][:, 0], vc_beta=d["vcap"][:, 1], vpw_alpha=d["vpw_alpha"][:, 0])
        for k in ("u_plunger_range", "u_plunger_center", "u_barrier_range", "u_barrier_center",
                  "u_start_plunger", "u_start_barrier"):
            out[k] = d[k]
        return out

    # -- a6: Maxwell matrices ---------------------------------------------------
    @staticmethod
    def maxwell(a):
        n, N = a["Cdd"].shape[0], a["Cdd"].shape[1]
        G = N + 1; nb = N - 1
        cdd_nm = np.zeros((n, G, G))
        cdd_nm[:, :N, :N] = a["Cdd"]
        cdd_nm[:, N:, :N] = a["Cds"]
        cdd_nm[:, :N, N:] = a["Cds"].transpose(0, 2, 1)
        cgd_nm = np.zeros((n, G, G + nb))
        cgd_nm[:, :N, :G] = a["Cgd"]
        cgd_nm[:, N:, :G] = a["Cgs"]
        cgd_nm[:, :N, G:] = a["Cbd"]
        cgd_nm[:, N:, G:] = a["Cbs"]
        cdd_sum = cdd_nm.sum(axis=2); cgd_sum = cgd_nm.sum(axis=2)
        off = cdd_nm.copy()
        off[:, np.arange(G), np.arange(G)] = 0.0
        cdd = -off
        cdd[:, np.arange(G), np.arange(G)] = cdd_sum + cgd_sum
        return cdd, np.linalg.inv(cdd), -cgd_nm

    def build(self, u) -> EpisodeBatch:
        """u: (n, n_draws) uniforms -> parameter blocks + initial state blocks."""
        N = self.N; G = N + 1; nb = N - 1; L = self.L
        a = self.assemble(self.draws_from_uniform(u))
        n = u.shape[0]
        cdd, cdd_inv, cgd = self.maxwell(a)
        eb = EpisodeBatch(n, N)
        P = eb.params
        P[:, L.cdd_inv:L.cdd_inv + G * G] = cdd_inv.reshape(n, -1)
        P[:, L.cgd:L.cgd + G * (G + nb)] = cgd.reshape(n, -1)
        P[:, L.cbg:L.cbg + nb * G] = a["Cbg"].reshape(n, -1)
        # A = U U^T with U upper triangular: Cholesky of the index-reversed matrix
        A = cdd_inv[:, :N, :N]
        Lr = np.linalg.cholesky(A[:, ::-1, ::-1])
        U = Lr[:, ::-1, ::-1]
        P[:, L.ufac:L.ufac + N * N] = U.reshape(n, -1)
        P[:, L.uinv:L.uinv + N] = 1.0 / U[:, np.arange(N), np.arange(N)]
        P[:, L.alpha:L.alpha + nb] = a["alpha"]
        origin = np.concatenate([a["offset"], np.zeros((n, 1))], axis=1)
        P[:, L.origin:L.origin + G] = origin
        # a21: optimal physical voltages, barrier targets
        Rm = np.linalg.cholesky(cdd_inv).transpose(0, 2, 1)
        M = np.linalg.pinv(Rm @ cgd[:, :, :G], rcond=1e-3) @ Rm
        vopt = np.einsum('nij,j->ni', M, self.n_star)
        tc_ratio = self.optimal_tc / a["tc_base"]
        vb_base = -np.log(tc_ratio)[:, None] / a["alpha"]
        vb_opt = vb_base - np.einsum('nbg,ng->nb', a["Cbg"], vopt)
        P[:, L.vopt:L.vopt + G] = vopt
        P[:, L.vbopt:L.vbopt + nb] = vb_opt
        # ground truth under the identity (electrons: -I) VGM, env.py:179-199
        if self.perfect_vgm:
            # qarray_base_class.py:879-901: -pinv(cdd_inv_full @ cgd_full[:, :n_gate]), negated for electrons
            vgm0 = np.linalg.pinv(cdd_inv @ cgd[:, :, :G])
        else:
            vgm0 = np.broadcast_to(-np.eye(G), (n, G, G))
        virt = np.linalg.solve(vgm0, (vopt - origin)[:, :, None])[:, :, 0]
        pgt = virt[:, :N].astype(np.float32); bgt = vb_opt.astype(np.float32); sgt = virt[:, N]
        # env.py:808-839 ranges, :842-858 start
        # NOTE the reference forms low/high in float32 (float32 ground truth combined with a
        # Python-float half width, env.py:819-822) before np.random.uniform widens them.
        pr = a["u_plunger_range"]
        h = (0.5 * (pr - 2)).astype(np.float32)
        lo = (pgt - h).astype(np.float64); hi = (pgt + h).astype(np.float64)
        pc = lo + (hi - lo) * a["u_plunger_center"]
        pmax = pc + 0.5 * pr; pmin = pc - 0.5 * pr
        br = a["u_barrier_range"]
        h = (0.5 * (br - 1)).astype(np.float32)
        lo = (bgt - h).astype(np.float64); hi = (bgt + h).astype(np.float64)
        bc = lo + (hi - lo) * a["u_barrier_center"]
        bmax = bc + 0.5 * br; bmin = bc - 0.5 * br
        P[:, L.pmin:L.pmin + N] = pmin; P[:, L.pmax:L.pmax + N] = pmax
        P[:, L.bmin:L.bmin + nb] = bmin; P[:, L.bmax:L.bmax + nb] = bmax
        P[:, L.scal + 0] = a["tc_base"]; P[:, L.scal + 1] = a["coulomb_peak_width"]
        P[:, L.scal + 2] = a["window_delta"]
        # f4 options.  Peak width: qarray_base_class.py:856-863 (the ctor override wins unless it is the 0.01 default)
        if self.vary_peak_width:
            P[:, L.scal + 3] = np.abs(self.peak_width_alpha) if self.peak_width_alpha != 0.01 else a["vpw_alpha"]
        else:
            P[:, L.scal + 3] = -1.0
        if self.vc_type == "linear":
            P[:, L.scal + 4] = 1.0; P[:, L.scal + 5] = a["vc_alpha"]; P[:, L.scal + 6] = a["vc_beta"]
        # a14 latching probabilities (qarray_base_class.py:495-519)
        P[:, L.pleads:L.pleads + N] = a["p_leads"]
        P[:, L.pinter:L.pinter + N * N] = a["p_inter"].reshape(n, -1)
        # a16 noise parameters (drawn in the reference order above; used only when the handle
        # is created with the corresponding noise flags)
        P[:, L.noise + 0] = a["white_noise_amplitude"]
        P[:, L.noise + 1] = a["telegraph"]["p01"]; P[:, L.noise + 2] = a["telegraph"]["p10"]
        P[:, L.noise + 3] = a["telegraph"]["amplitude"]
        rn = self.ecfg["simulator"].get("radial_noise") or {"enabled": False}
        if rn.get("enabled"):
            col = 0
            def take(v):
                nonlocal col
                if isinstance(v, dict):
                    out = a["radial"][:, col]; col += 1
                    return out
                return np.full(n, float(v))
            zr = take(rn["lower"]); dl = take(rn["ramp_range"])
            tn = rn.get("total_noise_range")
            full = take(tn) if isinstance(tn, dict) else np.full(n, -1.0)
            P[:, L.noise + 4] = zr; P[:, L.noise + 5] = zr + dl; P[:, L.noise + 6] = full
            P[:, L.noise + 7] = float(rn["max_amplitude"])
        else:
            P[:, L.noise + 6] = -1.0
        S = eb.state
        S[:, L.s_vgm:L.s_vgm + G * G] = vgm0.reshape(n, -1)
        S[:, L.s_gate_v:L.s_gate_v + N] = pmin + (pmax - pmin) * a["u_start_plunger"]
        S[:, L.s_barrier_v:L.s_barrier_v + nb] = bmin + (bmax - bmin) * a["u_start_barrier"]
        S[:, L.s_gate_gt:L.s_gate_gt + N] = pgt
        S[:, L.s_barrier_gt:L.s_barrier_gt + nb] = bgt
        S[:, L.s_sensor_gt] = sgt
        eb.extras = dict(a, cdd_inv=cdd_inv, cgd=cgd, vopt=vopt, vb_opt=vb_opt)
        return eb

    @property
    def n_draws(self):
        return len(self.plan.lo)


def check_solver_options(qconfig):
    """`simulator.latched_model` of qarray_config.yaml (qarray_config.yaml:127-130 in the reference) selects how the
    reference solves each pixel.  This library implements the default: 32 kept charge states and the EXACT ground state of
    their Hamiltonian (the reference: dense `jnp.linalg.eigh`, ground_state.py:149-162).  Two options would silently give
    other numbers than the reference and are refused instead:
      * `use_sparse: true` -- the reference then approximates the ground state by 50 float32 Lanczos steps from the uniform
        superposition at ONE representative tunnel coupling (fully_sparse_jax_eigensolver.py:68-133, ground_state.py:117-147);
        that approximation is not built;
      * `num_charge_states` other than 32 -- the kernels keep exactly 32 (QD_K).
    `charge_state_batch_size` only chunks the reference's scan and does not change its result (any value is fine)."""
    lm = ((qconfig or {}).get("simulator") or {}).get("latched_model") or {}
    if lm.get("use_sparse"):
        raise NotImplementedError("latched_model.use_sparse: true selects the reference's 50-step float32 Lanczos approximation "
                                  "(fully_sparse_jax_eigensolver.py:68-133); qadapt_hip only computes the exact ground state "
                                  "(the default, use_sparse: false)")
    k = lm.get("num_charge_states", 32)
    if k is not None and int(k) != 32:
        raise NotImplementedError(f"latched_model.num_charge_states = {k}: the kernels keep exactly 32 charge states")

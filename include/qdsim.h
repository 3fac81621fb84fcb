/*
 * qdsim.h -- C ABI of libqdsim.so, the MI355X-native batched replacement for the
 * per-step charge-stability simulation of QADAPT (edwindn/rl-agent-for-qubit-
 * array-tuning).  Plain C, plain pointers and sizes, no torch types; every
 * function returns 0 on success or a QD_ERR_* code (text via qd_last_error).
 *
 * The reference has no FFI (it is 100 % Python); each entry point below names
 * the Python interface it stands in for (paths relative to the reference root).
 * INTEGRATION.md shows the ctypes stub a maintainer adds on the reference side.
 *
 * Memory contract: all `*_dev` pointers are DEVICE memory owned by the caller
 * (e.g. torch.Tensor.data_ptr()) and must stay valid until the stream work that
 * uses them has finished.  `stream` is a hipStream_t passed as void* (NULL =
 * default stream).  A handle is bound to one GPU and is not thread-safe (the
 * RLlib env runner that calls the reference is single-threaded as well).
 *
 * Batch layout: B environments of the same n_dot (N) and resolution (R).
 *   C = N-1 CSD channels, P = R*R pixels, G = N+1 gates (plungers + sensor).
 *   agent order everywhere: plunger_0..plunger_{N-1}, barrier_0..barrier_{N-2}.
 */
#ifndef QDSIM_H
#define QDSIM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QD_OK 0
#define QD_ERR_ARG 1          /* bad argument / unsupported configuration        */
#define QD_ERR_HIP 2          /* a HIP runtime call failed                        */
#define QD_ERR_STATE 3        /* call order violated (e.g. outputs not bound)     */
#define QD_ERR_NOMEM 4

#define QD_FLAG_VALIDATE 1    /* keep per-pixel candidate records and occupations  */
#define QD_FLAG_PIXEL_SEARCH 2 /* a9 by the per-pixel search only (default: one search per 8x8 pixel tile where the
                                 grid is fine enough, with an exact per-pixel redo pass; same results, A/B switch) */
#define QD_FLAG_RETIRED_TILE_FUSED 4  /* was QD_FLAG_TILE_FUSED (round 2: experimental fused tile kernel, 3x slower than the
                                 default pipeline); the kernel is gone and qd_create refuses the flag with QD_ERR_ARG */

/* Stochastic stages.  The generators are counter-based Philox streams, so
 * results are reproducible per (rng_seed, global env id, observation number, channel, pixel)
 * but NOT comparable sample-by-sample with the reference's numpy global RNG. */
#define QD_NOISE_SENSOR 1     /* white + telegraph noise on the sensor potential
                                 (TunnelCoupledChargeSensed.py:354; qarray WhiteNoise/TelegraphNoise) */
#define QD_NOISE_RADIAL 2     /* distance-dependent image noise / white-noise replacement
                                 (qarray_base_class.py:444-493)                     */
#define QD_NOISE_LATCH 4      /* charge latching along the raster (ground_state.py:164; qarray
                                 LatchingModel, source absent: UNVERIFIED restatement)     */

typedef struct qd_handle qd_handle;

/* Mirrors the env_config.yaml / constructor knobs of QuantumDeviceEnv
 * (src/qadapt/environment/env.py:38-127, 350-462, 779-787). */
typedef struct qd_config {
    int32_t struct_size;          /* = sizeof(qd_config), for ABI checks          */
    int32_t n_dot;                /* N, 2..8                                       */
    int32_t resolution;           /* R (env_config.yaml simulator.resolution)     */
    int32_t batch;                /* B environments held by this handle            */
    int32_t max_steps;            /* truncation horizon (simulator.max_steps)     */
    int32_t env_chunk;            /* envs per scratch chunk, 0 = choose            */
    int32_t flags;                /* QD_FLAG_*                                     */
    int32_t noise_flags;          /* QD_NOISE_* (0 = deterministic parity mode)           */
    double gate_ramp_start;       /* reward.gate_ramp_start                        */
    double gate_quadratic_start;  /* reward.gate_quadratic_start                   */
    double barrier_ramp_start;    /* reward.barrier_ramp_start                     */
    double kalman_prior_mean;     /* env.py:781                                    */
    double kalman_prior_variance; /* env.py:782                                    */
    double kalman_prior_mean_nnn; /* env.py:786                                    */
    double kalman_variance_threshold; /* capacitance_model.variance_threshold     */
    double kalman_process_noise;  /* capacitance_model.process_noise               */
    uint64_t rng_seed;            /* Philox key for the stochastic stages; the two 32-bit halves are XOR-folded into ONE
                                     32-bit key word (seeds that differ only by such a fold give the same streams)      */
    int64_t env_id_offset;        /* global id of env 0 (multi-GPU shards); the second key word is the low 32 bits of
                                     env_id_offset + env index: global env ids are taken modulo 2^32                    */
    /* config variants reachable from env_config.yaml (env.py:393-441, 553-563, 592-618, 861-876) */
    int32_t use_deltas;           /* simulator.use_deltas: gate actions are increments (env.py:864-867) */
    int32_t sparse_reward;        /* reward.sparse_reward (env.py:393-414)         */
    int32_t gate_curve_type;      /* QD_CURVE_* (reward.gate_curve_type, env.py:430-441) */
    int32_t update_method;        /* QD_UPDATE_KALMAN / QD_UPDATE_DIRECT (KalmanUpdater.py / DirectUpdater.py) */
    int32_t cnn_outputs;          /* 3: [NN, NNN_right, NNN_left]; 2: legacy nearest_neighbour [RL, LR] (env.py:592-618) */
    int32_t reserved0;
    double delta_max;             /* simulator.delta_max                           */
    double gate_curve_exponent;   /* reward.gate_curve_exponent                    */
    double plunger_radius;        /* reward.plunger_radius        (sparse)         */
    double outer_plunger_radius;  /* reward.outer_plunger_radius  (sparse)         */
    double outer_plunger_reward_max; /* reward.outer_plunger_reward_max (sparse)  */
    double barrier_radius;        /* reward.barrier_radius        (sparse)         */
} qd_config;

#define QD_CURVE_CONSTANT 0
#define QD_CURVE_POLYNOMIAL 1
#define QD_CURVE_EXPONENTIAL 2
#define QD_CURVE_LINEAR 3
#define QD_UPDATE_KALMAN 0
#define QD_UPDATE_DIRECT 1

/* Sizes (in float64 elements) of the per-env parameter and state blocks whose
 * layout is documented in csrc/qd_common.h (mirrored by qadapt_hip/layout.py). */
int qd_param_block_doubles(int n_dot);
int qd_state_block_doubles(int n_dot);
/* Writes the 31 layout integers (see qadapt_hip/layout.py LAYOUT_FIELDS). */
int qd_layout_query(int n_dot, int32_t* out31);

/* QuantumDeviceEnv.__init__ (env.py:38-132): allocates device state for B envs
 * on GPU `device`; Kalman filters start at their priors (env.py:779-787). */
/* On failure after the handle was allocated (QD_ERR_HIP / QD_ERR_NOMEM), *out is still set: read the message with
 * qd_last_error and release the partial handle with qd_destroy. */
int qd_create(const qd_config* cfg, int device, qd_handle** out);
int qd_destroy(qd_handle* h);
const char* qd_last_error(const qd_handle* h);

/* Output tensors written by qd_observe (all float32, caller-owned device memory):
 *   global_image   [B][R][R][C]    obs["image"]            (env.py:471-509)
 *   plunger_images [B][N][R][R][2] per-agent plunger view  (multi_agent_wrapper.py:311-350)
 *   barrier_images [B][C][R][R][1] per-agent barrier view
 *   voltages       [B][2N-1]       normalised gate then barrier voltages (env.py:511-532)
 * Any pointer may be NULL to skip that output. */
int qd_bind_outputs(qd_handle* h, float* global_image_dev, float* plunger_images_dev,
                    float* barrier_images_dev, float* voltages_dev);

/* QuantumDeviceEnv.reset's device construction (env.py:160-222) for `n` envs:
 * uploads parameter blocks and initial state blocks (HOST pointers, n rows each)
 * built by the host-side sampler; step counters return to 0.  The Kalman state
 * is NOT touched (the reference never resets it, env.py:130) unless
 * reset_kalman != 0. */
int qd_load_episodes(qd_handle* h, const int32_t* env_ids_host, int n, const double* params_host,
                     const double* state_host, int reset_kalman, void* stream);

/* QuantumDeviceEnv.step lines env.py:260-285: clip + rescale the actions
 * [B][2N-1] (gates then barriers, float32), reward against the PREVIOUS ground
 * truth, step counter and truncation flag.  rewards_dev [B][2N-1] float64,
 * truncated_dev [B] uint8. */
int qd_apply_actions(qd_handle* h, const float* actions_dev, double* rewards_dev,
                     uint8_t* truncated_dev, void* stream);

/* QarrayBaseClass._get_obs + QuantumDeviceEnv._normalise_obs +
 * MultiAgentEnvWrapper._extract_agent_observation
 * (qarray_base_class.py:171-229, env.py:471-534, multi_agent_wrapper.py:311-383)
 * for the listed envs (env_ids_dev == NULL: all B).  Writes the bound outputs. */
int qd_observe(qd_handle* h, const int32_t* env_ids_dev, int n, void* stream);

/* QuantumDeviceEnv._update_virtual_gate_matrix (env.py:537-622) with the CNN's
 * outputs supplied by the caller: values/log_vars [B][C][3] float32 (indexed by
 * env id, not by list position).  Kalman update (KalmanUpdater.py:92-213), VGM
 * (qarray_base_class.py:904-942) and, if recompute_ground_truth != 0, the new
 * ground truth (env.py:298-305; reset() skips this, env.py:233). */
int qd_update_capacitance(qd_handle* h, const int32_t* env_ids_dev, int n, const float* values_dev,
                          const float* log_vars_dev, int recompute_ground_truth, void* stream);

/* One whole QuantumDeviceEnv.step for all B envs = qd_apply_actions +
 * qd_observe + qd_update_capacitance(recompute_ground_truth = 1). */
int qd_step(qd_handle* h, const float* actions_dev, const float* values_dev,
            const float* log_vars_dev, double* rewards_dev, uint8_t* truncated_dev, void* stream);

/* Validation / checkpoint access (blocking copies to HOST memory).
 *   state_host [B][qd_state_block_doubles], steps_host [B] int32
 *   raw_host   [B][C][P] float64 unnormalised sensor signal of the last observe
 *   plohi_host [B][2]    the 0.5 / 99.5 percentiles used
 *   occ_host   [B][C][P][N] float64 expectation occupations   (QD_FLAG_VALIDATE)
 *   states_host [B][C][P][32][N] int32 kept charge states      (QD_FLAG_VALIDATE) */
int qd_get_state(qd_handle* h, double* state_host, int32_t* steps_host);
int qd_set_state(qd_handle* h, const double* state_host, const int32_t* steps_host);
int qd_get_raw(qd_handle* h, double* raw_host, double* plohi_host);
int qd_get_occupations(qd_handle* h, double* occ_host);
int qd_get_candidates(qd_handle* h, int32_t* states_host);
/* eig_host [B][C][P][2] float64 (QD_FLAG_VALIDATE): per pixel the ground energy of the 32-state
 * Hamiltonian (what jnp.linalg.eigh returns first, ground_state.py:150) and the relative residual
 * ||H x - lambda x||_2 / ||H||_inf of the eigenpair the occupations were formed from. */
int qd_get_eigen(qd_handle* h, double* eig_host);
/* Counters of the tile-shared candidate search since qd_create (QD_FLAG_VALIDATE): tiles searched, tiles handed
 * whole to the per-pixel search, single pixels redone, sum of superset sizes, pixels redone for < 32 valid states;
 * [8 + r]: tiles handed over by reason r (1 ranges, 2 seeds, 3 frontier overflow, 4 too few leaves, 5 superset size). */
int qd_get_search_stats(qd_handle* h, uint64_t* out16);
/* Counters of the ground-state kernel's eigen-solver phase since qd_create (QD_FLAG_VALIDATE): tasks (hop components of
 * >= 2 states solved), sum of their Laguerre iterations, 64-task wave tiles, sum over tiles of the largest iteration count
 * in the tile; [4 + k]: tasks of size class k (2, 3, .. 8 states, then 9..32). */
int qd_get_solver_stats(qd_handle* h, uint64_t* out16);
/* Checkpointing of the stochastic stages (SURVEY 5 "expose RNG seeds/counters"): the Philox
 * counter word that numbers the observations rendered so far by this handle. */
int qd_get_rng_state(const qd_handle* h, uint64_t* obs_serial);
int qd_set_rng_state(qd_handle* h, uint64_t obs_serial);

/* Timing hooks for bench.py: `iters` back-to-back launches over ONE launch chunk (qd_chunk_envs envs) on the current
 * data, mean duration in milliseconds measured with HIP events on `stream`.
 *   qd_time_candidates_kernel  the whole candidate search (tile search + per-pixel redo pass)
 *   qd_time_ground_kernel      the whole ground-state stage (structure + the solve launches + select)
 *   qd_time_kernels            each kernel (group) by itself; out[QD_TIMED_KERNELS] in the order of qd_timed_kernel_name:
 *                              tile search, per-pixel redo pass, ground-state structure, the solve launches of all size
 *                              classes together, ground-state select. */
#define QD_TIMED_KERNELS 5
int qd_time_ground_kernel(qd_handle* h, int iters, float* mean_ms, void* stream);
int qd_time_candidates_kernel(qd_handle* h, int iters, float* mean_ms, void* stream);
int qd_time_kernels(qd_handle* h, int iters, float* mean_ms_out, void* stream);
const char* qd_timed_kernel_name(int k);
/* Number of env-steps one launch of the hot kernels covers (scratch chunk). */
int qd_chunk_envs(const qd_handle* h);

#ifdef __cplusplus
}
#endif
#endif /* QDSIM_H */

// qd_api.hip -- C-ABI of libqdsim.so (include/qdsim.h): handle, device buffers,
// kernel launches.  gfx950 only; no torch, no exceptions across the boundary.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>

#include "qdsim.h"
#include "qd_kernels.h"

struct qd_handle {
    qd_config cfg;
    int device;
    QdLayout L;
    int N, R, B, C, P;
    int chunk;
    double *params, *state, *zraw, *plohi, *occ;
    int* steps;
    QdPixelRec* recs;
    size_t recs_envs;                       // envs the recs buffer holds
    float *gimg, *pimg, *bimg, *volt;
    unsigned long long* tel; int tel_words;
    unsigned long long obs_serial;
    char err[512];
};

static int qd_fail(qd_handle* h, int code, const char* what, hipError_t e = hipSuccess) {
    if (h) {
        if (e != hipSuccess) snprintf(h->err, sizeof(h->err), "%s: %s", what, hipGetErrorString(e));
        else snprintf(h->err, sizeof(h->err), "%s", what);
    }
    return code;
}
#define QD_HIP(call)                                                              \
    do { hipError_t e_ = (call); if (e_ != hipSuccess) return qd_fail(h, QD_ERR_HIP, #call, e_); } while (0)

#define QD_DISPATCH_N(N_, ...)                                                    \
    switch (N_) {                                                                 \
        case 2: { constexpr int NN = 2; __VA_ARGS__; } break;                            \
        case 3: { constexpr int NN = 3; __VA_ARGS__; } break;                            \
        case 4: { constexpr int NN = 4; __VA_ARGS__; } break;                            \
        case 5: { constexpr int NN = 5; __VA_ARGS__; } break;                            \
        case 6: { constexpr int NN = 6; __VA_ARGS__; } break;                            \
        case 7: { constexpr int NN = 7; __VA_ARGS__; } break;                            \
        case 8: { constexpr int NN = 8; __VA_ARGS__; } break;                            \
        default: return qd_fail(h, QD_ERR_ARG, "n_dot must be in 2..8");          \
    }

extern "C" int qd_param_block_doubles(int n) { return (n < 2 || n > QD_MAXN) ? -1 : qd_layout(n).size; }
extern "C" int qd_state_block_doubles(int n) { return (n < 2 || n > QD_MAXN) ? -1 : qd_layout(n).s_size; }
extern "C" int qd_layout_query(int n, int32_t* out) {
    if (n < 2 || n > QD_MAXN || !out) return QD_ERR_ARG;
    QdLayout L = qd_layout(n);
    memcpy(out, &L, sizeof(L));
    return QD_OK;
}

extern "C" const char* qd_last_error(const qd_handle* h) { return h ? h->err : "null handle"; }

extern "C" int qd_create(const qd_config* cfg, int device, qd_handle** out) {
    if (!cfg || !out || cfg->struct_size != (int32_t)sizeof(qd_config)) return QD_ERR_ARG;
    if (cfg->n_dot < 2 || cfg->n_dot > QD_MAXN || cfg->resolution < 2 || cfg->batch < 1) return QD_ERR_ARG;
    qd_handle* h = new (std::nothrow) qd_handle();
    if (!h) return QD_ERR_NOMEM;
    memset(h, 0, sizeof(*h));
    h->cfg = *cfg; h->device = device;
    h->N = cfg->n_dot; h->R = cfg->resolution; h->B = cfg->batch;
    h->C = h->N - 1; h->P = h->R * h->R; h->L = qd_layout(h->N);
    *out = h;
    QD_HIP(hipSetDevice(device));
    const size_t per_env_rec = (size_t)h->C * h->P * sizeof(QdPixelRec);
    int chunk = cfg->env_chunk;
    if (cfg->flags & QD_FLAG_VALIDATE) chunk = h->B;
    else if (chunk <= 0) {
        const size_t budget = (size_t)1 << 30;                 // 1 GiB of candidate records in flight
        chunk = (int)(budget / per_env_rec);
        if (chunk < 1) chunk = 1;
    }
    if (chunk > h->B) chunk = h->B;
    h->chunk = chunk; h->recs_envs = (size_t)chunk;
    QD_HIP(hipMalloc(&h->params, sizeof(double) * (size_t)h->B * h->L.size));
    QD_HIP(hipMalloc(&h->state, sizeof(double) * (size_t)h->B * h->L.s_size));
    QD_HIP(hipMalloc(&h->steps, sizeof(int) * (size_t)h->B));
    QD_HIP(hipMalloc(&h->zraw, sizeof(double) * (size_t)h->B * h->C * h->P));
    QD_HIP(hipMalloc(&h->plohi, sizeof(double) * 2 * (size_t)h->B));
    QD_HIP(hipMalloc(&h->recs, per_env_rec * h->recs_envs));
    h->tel_words = (h->P + 63) / 64;
    if (cfg->noise_flags & QD_NOISE_SENSOR) {
        QD_HIP(hipMalloc(&h->tel, sizeof(unsigned long long) * (size_t)h->B * h->C * h->tel_words));
        QD_HIP(hipMemset(h->tel, 0, sizeof(unsigned long long) * (size_t)h->B * h->C * h->tel_words));
    }
    if ((cfg->flags & QD_FLAG_VALIDATE) || (cfg->noise_flags & QD_NOISE_LATCH))
        QD_HIP(hipMalloc(&h->occ, sizeof(double) * (size_t)h->B * h->C * h->P * h->N));
    QD_HIP(hipMemset(h->params, 0, sizeof(double) * (size_t)h->B * h->L.size));
    QD_HIP(hipMemset(h->steps, 0, sizeof(int) * (size_t)h->B));
    QD_HIP(hipMemset(h->zraw, 0, sizeof(double) * (size_t)h->B * h->C * h->P));
    QD_HIP(hipMemset(h->plohi, 0, sizeof(double) * 2 * (size_t)h->B));
    // Kalman priors (KalmanUpdater.py:64-81) into every env's state block
    {
        const int N = h->N;
        double* host = (double*)calloc((size_t)h->B * h->L.s_size, sizeof(double));
        if (!host) return qd_fail(h, QD_ERR_NOMEM, "calloc");
        for (int e = 0; e < h->B; ++e) {
            double* st = host + (size_t)e * h->L.s_size;
            for (int i = 0; i < N + 1; ++i) st[h->L.s_vgm + i * (N + 1) + i] = -1.0;
            for (int i = 0; i < N - 1; ++i) {
                st[h->L.s_kmean + i * N + i + 1] = st[h->L.s_kmean + (i + 1) * N + i] = cfg->kalman_prior_mean;
                st[h->L.s_kvar + i * N + i + 1] = st[h->L.s_kvar + (i + 1) * N + i] = cfg->kalman_prior_variance;
            }
            for (int i = 0; i < N - 2; ++i) {
                st[h->L.s_kmean + i * N + i + 2] = st[h->L.s_kmean + (i + 2) * N + i] = cfg->kalman_prior_mean_nnn;
                st[h->L.s_kvar + i * N + i + 2] = st[h->L.s_kvar + (i + 2) * N + i] = cfg->kalman_prior_variance;
            }
        }
        hipError_t e_ = hipMemcpy(h->state, host, sizeof(double) * (size_t)h->B * h->L.s_size, hipMemcpyHostToDevice);
        free(host);
        if (e_ != hipSuccess) return qd_fail(h, QD_ERR_HIP, "hipMemcpy(state)", e_);
    }
    snprintf(h->err, sizeof(h->err), "ok");
    return QD_OK;
}

extern "C" int qd_destroy(qd_handle* h) {
    if (!h) return QD_ERR_ARG;
    (void)hipSetDevice(h->device);
    void* bufs[] = {h->params, h->state, h->steps, h->zraw, h->plohi, h->recs, h->occ, h->tel};
    for (void* b : bufs) if (b) (void)hipFree(b);
    delete h;
    return QD_OK;
}

extern "C" int qd_bind_outputs(qd_handle* h, float* g, float* p, float* b, float* v) {
    if (!h) return QD_ERR_ARG;
    h->gimg = g; h->pimg = p; h->bimg = b; h->volt = v;
    return QD_OK;
}

extern "C" int qd_load_episodes(qd_handle* h, const int32_t* env_ids, int n, const double* params,
                                const double* state, int reset_kalman, void* stream) {
    if (!h || !env_ids || n < 0 || !params || !state) return qd_fail(h, QD_ERR_ARG, "qd_load_episodes: bad argument");
    hipStream_t s = (hipStream_t)stream;
    QD_HIP(hipSetDevice(h->device));
    const QdLayout& L = h->L;
    const int N = h->N;
    const size_t pre = (size_t)L.s_kmean;                        // everything before the Kalman block
    bool contiguous = n > 1 && !reset_kalman;
    for (int k = 1; k < n && contiguous; ++k) contiguous = env_ids[k] == env_ids[0] + k;
    if (contiguous) {
        const int e0 = env_ids[0];
        if (e0 < 0 || e0 + n > h->B) return qd_fail(h, QD_ERR_ARG, "qd_load_episodes: env id out of range");
        QD_HIP(hipMemcpyAsync(h->params + (size_t)e0 * L.size, params, sizeof(double) * L.size * n,
                              hipMemcpyHostToDevice, s));
        QD_HIP(hipMemcpy2DAsync(h->state + (size_t)e0 * L.s_size, sizeof(double) * L.s_size, state,
                                sizeof(double) * L.s_size, sizeof(double) * pre, n, hipMemcpyHostToDevice, s));
        QD_HIP(hipMemsetAsync(h->steps + e0, 0, sizeof(int) * n, s));
        QD_HIP(hipStreamSynchronize(s));
        return QD_OK;
    }
    for (int k = 0; k < n; ++k) {
        const int e = env_ids[k];
        if (e < 0 || e >= h->B) return qd_fail(h, QD_ERR_ARG, "qd_load_episodes: env id out of range");
        QD_HIP(hipMemcpyAsync(h->params + (size_t)e * L.size, params + (size_t)k * L.size,
                              sizeof(double) * L.size, hipMemcpyHostToDevice, s));
        QD_HIP(hipMemcpyAsync(h->state + (size_t)e * L.s_size, state + (size_t)k * L.s_size,
                              sizeof(double) * pre, hipMemcpyHostToDevice, s));
        if (reset_kalman) {
            double kal[2 * QD_MAXN * QD_MAXN];
            memset(kal, 0, sizeof(kal));
            double* km = kal; double* kv = kal + N * N;
            for (int i = 0; i < N - 1; ++i) {
                km[i * N + i + 1] = km[(i + 1) * N + i] = h->cfg.kalman_prior_mean;
                kv[i * N + i + 1] = kv[(i + 1) * N + i] = h->cfg.kalman_prior_variance;
            }
            for (int i = 0; i < N - 2; ++i) {
                km[i * N + i + 2] = km[(i + 2) * N + i] = h->cfg.kalman_prior_mean_nnn;
                kv[i * N + i + 2] = kv[(i + 2) * N + i] = h->cfg.kalman_prior_variance;
            }
            QD_HIP(hipMemcpy(h->state + (size_t)e * L.s_size + L.s_kmean, kal, sizeof(double) * 2 * N * N,
                             hipMemcpyHostToDevice));
        }
        QD_HIP(hipMemsetAsync(h->steps + e, 0, sizeof(int), s));
    }
    // pageable host memory: make sure the copies have consumed the caller's buffers
    QD_HIP(hipStreamSynchronize(s));
    return QD_OK;
}

extern "C" int qd_apply_actions(qd_handle* h, const float* actions, double* rewards, uint8_t* truncated, void* stream) {
    if (!h || !actions) return qd_fail(h, QD_ERR_ARG, "qd_apply_actions: bad argument");
    hipStream_t s = (hipStream_t)stream;
    QD_HIP(hipSetDevice(h->device));
    QdRewardCfg rc{h->cfg.gate_ramp_start, h->cfg.gate_quadratic_start, h->cfg.barrier_ramp_start, h->cfg.max_steps};
    const int blk = 64, grd = (h->B + blk - 1) / blk;
    QD_DISPATCH_N(h->N, qd_k_actions<NN><<<dim3(grd), dim3(blk), 0, s>>>(h->B, h->params,
                                            h->state, h->steps, actions, rewards, truncated, rc));
    QD_HIP(hipGetLastError());
    return QD_OK;
}

static int qd_cand_blocks(int R) {
    const int tiles = ((R + 7) / 8) * ((R + 7) / 8), per_block = QD_CAND_BLOCK / 64;
    return (tiles + per_block - 1) / per_block;
}

static QdNoiseCfg qd_noise_cfg(const qd_handle* h) {
    QdNoiseCfg nz;
    nz.flags = h->cfg.noise_flags;
    nz.seed = (uint32_t)(h->cfg.rng_seed ^ (h->cfg.rng_seed >> 32));
    nz.env_off = (uint32_t)h->cfg.env_id_offset;
    nz.ser_lo = (uint32_t)h->obs_serial; nz.ser_hi = (uint32_t)(h->obs_serial >> 32);
    nz.tel = h->tel; nz.tel_words = h->tel_words;
    return nz;
}

static int qd_launch_ground(qd_handle* h, const int32_t* env_ids, int base, int cnt, hipStream_t s) {
    dim3 g2((h->P + QD_GS_PPB - 1) / QD_GS_PPB, h->C, cnt);
    QD_DISPATCH_N(h->N, qd_k_ground<NN><<<g2, dim3(QD_GS_BLOCK), 0, s>>>(env_ids, base, h->R,
                                            h->params, h->recs, h->zraw, h->occ, h->state, h->cfg.noise_flags));
    QD_HIP(hipGetLastError());
    return QD_OK;
}

extern "C" int qd_observe(qd_handle* h, const int32_t* env_ids, int n, void* stream) {
    if (!h || n < 0) return qd_fail(h, QD_ERR_ARG, "qd_observe: bad argument");
    hipStream_t s = (hipStream_t)stream;
    QD_HIP(hipSetDevice(h->device));
    if (!env_ids) n = h->B;
    if (n == 0) return QD_OK;
    if (n > h->B) return qd_fail(h, QD_ERR_ARG, "qd_observe: n > batch");
    const QdLayout& L = h->L;
    const size_t shm = (size_t)QD_K * QD_CAND_BLOCK * (sizeof(double) + sizeof(uint16_t));
    h->obs_serial++;
    if (h->cfg.noise_flags & QD_NOISE_SENSOR) {
        const int nt = n * h->C;
        qd_k_telegraph<<<dim3((nt + 63) / 64), dim3(64), 0, s>>>(env_ids, n, h->C, h->P, L.size, L.noise, h->params, h->tel, qd_noise_cfg(h));
        QD_HIP(hipGetLastError());
    }
    for (int base = 0; base < n; base += h->chunk) {
        const int cnt = (n - base < h->chunk) ? n - base : h->chunk;
        dim3 g1(qd_cand_blocks(h->R), h->C, cnt);
        QD_DISPATCH_N(h->N, qd_k_candidates<NN><<<g1, dim3(QD_CAND_BLOCK), shm, s>>>(env_ids, base, h->R, h->params, h->state, h->recs, (h->cfg.flags & QD_FLAG_VALIDATE) ? 1 : 0, h->cfg.noise_flags));
        QD_HIP(hipGetLastError());
        int rc = qd_launch_ground(h, env_ids, base, cnt, s);
        if (rc) return rc;
    }
    if (h->cfg.noise_flags & QD_NOISE_LATCH) {
        const int nt = n * h->C;
        QD_DISPATCH_N(h->N, qd_k_latch<NN><<<dim3((nt + 63) / 64), dim3(64), 0, s>>>(env_ids, n, h->R, h->params, h->state, h->occ, h->zraw, qd_noise_cfg(h)));
        QD_HIP(hipGetLastError());
    }
    {
        dim3 g3((h->P + 255) / 256, h->C, n);
        QD_DISPATCH_N(h->N, qd_k_sensor<NN><<<g3, dim3(256), 0, s>>>(env_ids, h->R, h->params, h->state, h->zraw, qd_noise_cfg(h)));
        QD_HIP(hipGetLastError());
    }
    qd_k_percentile<<<dim3(n), dim3(QD_PCT_BLOCK), 0, s>>>(env_ids, (long)h->C * h->P, h->zraw, h->plohi);
    QD_HIP(hipGetLastError());
    if (h->gimg || h->pimg || h->bimg || h->volt) {
        dim3 g4((h->P + 255) / 256, n);
        QD_DISPATCH_N(h->N, qd_k_write_obs<NN><<<g4, dim3(256), 0, s>>>(env_ids, h->R, h->params,
                                                h->state, h->zraw, h->plohi, h->gimg, h->pimg, h->bimg, h->volt));
        QD_HIP(hipGetLastError());
    }
    return QD_OK;
}

extern "C" int qd_update_capacitance(qd_handle* h, const int32_t* env_ids, int n, const float* values,
                                     const float* log_vars, int recompute_gt, void* stream) {
    if (!h || n < 0) return qd_fail(h, QD_ERR_ARG, "qd_update_capacitance: bad argument");
    hipStream_t s = (hipStream_t)stream;
    QD_HIP(hipSetDevice(h->device));
    if (!env_ids) n = h->B;
    if (n == 0) return QD_OK;
    QdKalmanCfg kc{h->cfg.kalman_variance_threshold, h->cfg.kalman_process_noise};
    const int blk = 64, grd = (n + blk - 1) / blk;
    QD_DISPATCH_N(h->N, qd_k_update<NN><<<dim3(grd), dim3(blk), 0, s>>>(env_ids, n, h->params,
                                            h->state, values, log_vars, recompute_gt, kc));
    QD_HIP(hipGetLastError());
    return QD_OK;
}

extern "C" int qd_step(qd_handle* h, const float* actions, const float* values, const float* log_vars,
                       double* rewards, uint8_t* truncated, void* stream) {
    int rc = qd_apply_actions(h, actions, rewards, truncated, stream);
    if (rc) return rc;
    rc = qd_observe(h, nullptr, 0, stream);
    if (rc) return rc;
    return qd_update_capacitance(h, nullptr, 0, values, log_vars, 1, stream);
}

extern "C" int qd_get_state(qd_handle* h, double* state, int32_t* steps) {
    if (!h) return QD_ERR_ARG;
    QD_HIP(hipSetDevice(h->device));
    QD_HIP(hipDeviceSynchronize());
    if (state) QD_HIP(hipMemcpy(state, h->state, sizeof(double) * (size_t)h->B * h->L.s_size, hipMemcpyDeviceToHost));
    if (steps) QD_HIP(hipMemcpy(steps, h->steps, sizeof(int) * (size_t)h->B, hipMemcpyDeviceToHost));
    return QD_OK;
}
extern "C" int qd_set_state(qd_handle* h, const double* state, const int32_t* steps) {
    if (!h) return QD_ERR_ARG;
    QD_HIP(hipSetDevice(h->device));
    QD_HIP(hipDeviceSynchronize());
    if (state) QD_HIP(hipMemcpy(h->state, state, sizeof(double) * (size_t)h->B * h->L.s_size, hipMemcpyHostToDevice));
    if (steps) QD_HIP(hipMemcpy(h->steps, steps, sizeof(int) * (size_t)h->B, hipMemcpyHostToDevice));
    return QD_OK;
}
extern "C" int qd_get_raw(qd_handle* h, double* raw, double* plohi) {
    if (!h) return QD_ERR_ARG;
    QD_HIP(hipSetDevice(h->device));
    QD_HIP(hipDeviceSynchronize());
    if (raw) QD_HIP(hipMemcpy(raw, h->zraw, sizeof(double) * (size_t)h->B * h->C * h->P, hipMemcpyDeviceToHost));
    if (plohi) QD_HIP(hipMemcpy(plohi, h->plohi, sizeof(double) * 2 * (size_t)h->B, hipMemcpyDeviceToHost));
    return QD_OK;
}
extern "C" int qd_get_occupations(qd_handle* h, double* occ) {
    if (!h || !occ) return QD_ERR_ARG;
    if (!h->occ) return qd_fail(h, QD_ERR_STATE, "qd_get_occupations needs QD_FLAG_VALIDATE");
    QD_HIP(hipSetDevice(h->device));
    QD_HIP(hipDeviceSynchronize());
    QD_HIP(hipMemcpy(occ, h->occ, sizeof(double) * (size_t)h->B * h->C * h->P * h->N, hipMemcpyDeviceToHost));
    return QD_OK;
}
extern "C" int qd_get_candidates(qd_handle* h, int32_t* states) {
    if (!h || !states) return QD_ERR_ARG;
    if (!(h->cfg.flags & QD_FLAG_VALIDATE)) return qd_fail(h, QD_ERR_STATE, "qd_get_candidates needs QD_FLAG_VALIDATE");
    QD_HIP(hipSetDevice(h->device));
    QD_HIP(hipDeviceSynchronize());
    const size_t nrec = (size_t)h->B * h->C * h->P;
    QdPixelRec* host = (QdPixelRec*)malloc(nrec * sizeof(QdPixelRec));
    if (!host) return qd_fail(h, QD_ERR_NOMEM, "malloc");
    hipError_t e_ = hipMemcpy(host, h->recs, nrec * sizeof(QdPixelRec), hipMemcpyDeviceToHost);
    if (e_ != hipSuccess) { free(host); return qd_fail(h, QD_ERR_HIP, "hipMemcpy(recs)", e_); }
    static const int DELTA[4] = {-1, 0, 1, 2};
    const int N = h->N;
    for (size_t r = 0; r < nrec; ++r)
        for (int m = 0; m < QD_K; ++m)
            for (int i = 0; i < N; ++i) {
                const int dig = (host[r].idx[m] >> (2 * (N - 1 - i))) & 3;
                states[(r * QD_K + m) * N + i] = m < host[r].nvalid ? host[r].fl[i] + DELTA[dig] : 0;
            }
    free(host);
    return QD_OK;
}

extern "C" int qd_time_ground_kernel(qd_handle* h, int iters, float* mean_ms, void* stream) {
    if (!h || iters < 1 || !mean_ms) return QD_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    QD_HIP(hipSetDevice(h->device));
    hipEvent_t a, b;
    QD_HIP(hipEventCreate(&a)); QD_HIP(hipEventCreate(&b));
    const int cnt = h->chunk < h->B ? h->chunk : h->B;
    QD_HIP(hipEventRecord(a, s));
    for (int i = 0; i < iters; ++i) { int rc = qd_launch_ground(h, nullptr, 0, cnt, s); if (rc) return rc; }
    QD_HIP(hipEventRecord(b, s));
    QD_HIP(hipEventSynchronize(b));
    float ms = 0.f;
    QD_HIP(hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    *mean_ms = ms / iters;
    return QD_OK;
}

extern "C" int qd_time_candidates_kernel(qd_handle* h, int iters, float* mean_ms, void* stream) {
    if (!h || iters < 1 || !mean_ms) return QD_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    QD_HIP(hipSetDevice(h->device));
    hipEvent_t a, b;
    QD_HIP(hipEventCreate(&a)); QD_HIP(hipEventCreate(&b));
    const int cnt = h->chunk < h->B ? h->chunk : h->B;
    const size_t shm = (size_t)QD_K * QD_CAND_BLOCK * (sizeof(double) + sizeof(uint16_t));
    dim3 g1(qd_cand_blocks(h->R), h->C, cnt);
    QD_HIP(hipEventRecord(a, s));
    for (int i = 0; i < iters; ++i) {
        QD_DISPATCH_N(h->N, qd_k_candidates<NN><<<g1, dim3(QD_CAND_BLOCK), shm, s>>>(nullptr, 0, h->R, h->params, h->state, h->recs, (h->cfg.flags & QD_FLAG_VALIDATE) ? 1 : 0, h->cfg.noise_flags));
    }
    QD_HIP(hipGetLastError());
    QD_HIP(hipEventRecord(b, s));
    QD_HIP(hipEventSynchronize(b));
    float ms = 0.f;
    QD_HIP(hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    *mean_ms = ms / iters;
    return QD_OK;
}

extern "C" int qd_chunk_envs(const qd_handle* h) { return h ? h->chunk : -1; }

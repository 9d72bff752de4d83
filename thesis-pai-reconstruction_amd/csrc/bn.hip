// BatchNorm2d training/eval kernels (HBM-bound elementwise + per-channel reductions).
// Replaces aten::native_batch_norm / native_batch_norm_backward issued by nn.BatchNorm2d at
// models/pix2pix.py:70,106.  The batch statistics themselves come from the convolution
// epilogues as per-tile partial sums (see gg_*.hip); nothing here re-reads the activation to
// compute them.
#include "common.h"

constexpr int STAGE_ROWS = 64;        // (scratch rows behind the partials: pai_bn_stats_buffer_rows keeps the old 2 x 64)
constexpr int BIG_ROWS = 2048;        // more partial rows than this: chunk sums first
constexpr int BIG_CHUNKS = 256;

// ---- forward statistics --------------------------------------------------------------
// Partial rows of a big layer (32768 rows of 2 x 128 floats behind a 1 x 1 convolution of the residual U-Net at 512 x 512:
// 33 MB) summed into BIG_CHUNKS fp64 rows by the whole chip, coalesced -- bn_finalize_wide_k alone walks them with C / 8
// workgroups (16 CUs, 149 us).
__global__ __launch_bounds__(256) void bn_stats_chunks_k(const float* stats, int R, int C2, int per, double* out) {
    const int col = blockIdx.y * 256 + threadIdx.x;
    if (col >= C2) return;
    const int r0 = blockIdx.x * per, r1 = min(R, r0 + per);
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int r = r0;
    for (; r + 3 < r1; r += 4) {
        const float a = stats[(size_t)r * C2 + col], b = stats[(size_t)(r + 1) * C2 + col];
        const float c = stats[(size_t)(r + 2) * C2 + col], d = stats[(size_t)(r + 3) * C2 + col];
        s0 += (double)a; s1 += (double)b; s2 += (double)c; s3 += (double)d;
    }
    for (; r < r1; ++r) s0 += (double)stats[(size_t)r * C2 + col];
    out[(size_t)blockIdx.x * C2 + col] = (s0 + s1) + (s2 + s3);
}

__global__ void bn_finalize_k(const float* stats, int R, int C, double count,
                              const float* gamma, const float* beta, float eps, float momentum,
                              int n_updates, float* running_mean, float* running_var,
                              int64_t* nbt, float* mean_o, float* rstd_o, float* scale_o, float* shift_o) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && nbt) *nbt += n_updates;
    if (c >= C) return;
    double s = 0.0, q = 0.0;
    for (int r = 0; r < R; ++r) {
        s += (double)stats[(size_t)r * 2 * C + c];
        q += (double)stats[(size_t)r * 2 * C + C + c];
    }
    const double mean = s / count;
    double var = q / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float sc = g * rstd;
    mean_o[c] = (float)mean;
    rstd_o[c] = rstd;
    scale_o[c] = sc;
    shift_o[c] = b - (float)mean * sc;
    if (running_mean && running_var) {
        const float unbiased = (float)(count > 1.0 ? var * count / (count - 1.0) : var);
        float rm = running_mean[c], rv = running_var[c];
        for (int u = 0; u < n_updates; ++u) {
            rm = (1.f - momentum) * rm + momentum * (float)mean;
            rv = (1.f - momentum) * rv + momentum * unbiased;
        }
        running_mean[c] = rm;
        running_var[c] = rv;
    }
}

// rows of the partial-sum buffer: the partials + scratch (BIG_CHUNKS fp64 rows for the chunk sums of a big layer)
extern "C" int pai_bn_stats_buffer_rows(int rows) { return rows + (rows > BIG_ROWS ? 2 * BIG_CHUNKS : 2 * STAGE_ROWS); }

// One-launch finalize for many partial rows: 8 channels x 128 row lanes per block sum the rows in fp64 (as the
// backward finalize does), then 8 threads turn the totals into mean / rstd / scale / shift and advance the running
// statistics.  Replaces the stage-1 + finalize pair (two dependent tiny launches per BatchNorm layer).
template <typename P>     // P: float partial rows, or the fp64 chunk sums of bn_stats_chunks_k
__global__ __launch_bounds__(1024) void bn_finalize_wide_k(const P* stats, int R, int C, double count,
                                                           const float* gamma, const float* beta, float eps,
                                                           float momentum, int n_updates, float* running_mean,
                                                           float* running_var, int64_t* nbt, float* mean_o,
                                                           float* rstd_o, float* scale_o, float* shift_o) {
    __shared__ double red[2][128][8];
    const int cl = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int c = blockIdx.x * 8 + cl;
    if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) *nbt += n_updates;
    double s = 0.0, q = 0.0;
    if (c < C) {
        int r = rl;
        for (; r + 7 * 128 < R; r += 8 * 128) {      // eight rows in flight per lane (see bn_bwd_finalize_k)
            P a[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a[u] = stats[(size_t)(r + u * 128) * 2 * C + c];
                b[u] = stats[(size_t)(r + u * 128) * 2 * C + C + c];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { s += (double)a[u]; q += (double)b[u]; }
        }
        for (; r < R; r += 128) {
            s += (double)stats[(size_t)r * 2 * C + c];
            q += (double)stats[(size_t)r * 2 * C + C + c];
        }
    }
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {        // 8 row lanes per wave by shuffles, 16 waves through LDS
        s += __shfl_xor(s, o, 64);
        q += __shfl_xor(q, o, 64);
    }
    if ((threadIdx.x & 63) < 8) {
        red[0][threadIdx.x >> 6][cl] = s;
        red[1][threadIdx.x >> 6][cl] = q;
    }
    __syncthreads();
    if (rl != 0 || c >= C) return;
    s = q = 0.0;
#pragma unroll
    for (int l = 0; l < 16; ++l) { s += red[0][l][cl]; q += red[1][l][cl]; }
    const double mean = s / count;
    double var = q / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float sc = g * rstd;
    mean_o[c] = (float)mean;
    rstd_o[c] = rstd;
    scale_o[c] = sc;
    shift_o[c] = b - (float)mean * sc;
    if (running_mean && running_var) {
        const float unbiased = (float)(count > 1.0 ? var * count / (count - 1.0) : var);
        float rm = running_mean[c], rv = running_var[c];
        for (int u = 0; u < n_updates; ++u) {
            rm = (1.f - momentum) * rm + momentum * (float)mean;
            rv = (1.f - momentum) * rv + momentum * unbiased;
        }
        running_mean[c] = rm;
        running_var[c] = rv;
    }
}

extern "C" int pai_bn_finalize(const float* stats, int rows, int C, int64_t count, const float* gamma,
                               const float* beta, float eps, float momentum, int n_updates,
                               float* running_mean, float* running_var, int64_t* num_batches_tracked,
                               float* mean, float* rstd, float* scale, float* shift, void* stream) {
    PAI_CHECK(stats && mean && rstd && scale && shift, "pai_bn_finalize: null pointer");
    PAI_CHECK(rows > 0 && C > 0 && count > 0, "pai_bn_finalize: bad sizes");
    hipStream_t s = (hipStream_t)stream;
    if (rows > BIG_ROWS) {
        // scratch = the 2 * BIG_CHUNKS float rows that follow the partials (pai_bn_stats_buffer_rows)
        double* chunks = (double*)(stats + (size_t)rows * 2 * C);
        const int per = cdiv(rows, BIG_CHUNKS);
        PAI_LAUNCH(bn_stats_chunks_k, dim3(BIG_CHUNKS, cdiv(2 * C, 256)), dim3(256), 0, s, stats, rows, 2 * C, per, chunks);
        PAI_LAUNCH_CHECK();
        PAI_LAUNCH(bn_finalize_wide_k<double>, dim3(cdiv(C, 8)), dim3(1024), 0, s, (const double*)chunks, cdiv(rows, per), C,
                   (double)count, gamma, beta, eps, momentum, n_updates, running_mean, running_var, num_batches_tracked, mean,
                   rstd, scale, shift);
        PAI_LAUNCH_CHECK();
        return 0;
    }
    if (rows > 16) {
        PAI_LAUNCH(bn_finalize_wide_k<float>, dim3(cdiv(C, 8)), dim3(1024), 0, s, stats, rows, C, (double)count, gamma,
                           beta, eps, momentum, n_updates, running_mean, running_var, num_batches_tracked, mean, rstd,
                           scale, shift);
        PAI_LAUNCH_CHECK();
        return 0;
    }
    PAI_LAUNCH(bn_finalize_k, dim3(cdiv(C, 64)), dim3(64), 0, s, stats, rows, C,
                       (double)count, gamma, beta, eps, momentum, n_updates, running_mean, running_var,
                       num_batches_tracked, mean, rstd, scale, shift);
    PAI_LAUNCH_CHECK();
    return 0;
}

__global__ void bn_eval_coeffs_k(int C, const float* gamma, const float* beta, const float* rm,
                                 const float* rv, float eps, float* scale, float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float rstd = 1.f / sqrtf(rv[c] + eps);
    const float sc = (gamma ? gamma[c] : 1.f) * rstd;
    scale[c] = sc;
    shift[c] = (beta ? beta[c] : 0.f) - rm[c] * sc;
}

extern "C" int pai_bn_eval_coeffs(int C, const float* gamma, const float* beta, const float* running_mean,
                                  const float* running_var, float eps, float* scale, float* shift,
                                  void* stream) {
    PAI_CHECK(running_mean && running_var && scale && shift, "pai_bn_eval_coeffs: null pointer");
    PAI_LAUNCH(bn_eval_coeffs_k, dim3(cdiv(C, 64)), dim3(64), 0, (hipStream_t)stream, C, gamma,
                       beta, running_mean, running_var, eps, scale, shift);
    PAI_LAUNCH_CHECK();
    return 0;
}

// ---- forward apply -------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_k(const T* z, int64_t nvec, int C, const float* scale,
                                                  const float* shift, int act, T* out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        const int c0 = (int)((i * 8) % C);
        float v[8], sc[8], sh[8];
        V8<T>::ld(z + i * 8, v);
        V8<float>::ld(scale + c0, sc);
        V8<float>::ld(shift + c0, sh);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = act_apply(fmaf(v[k], sc[k], sh[k]), act);
        V8<T>::st(out + i * 8, v);
    }
}

// (two 16-B vectors in flight per thread in bn_apply_k / bn_bwd_apply_k were measured in round 3: no change of the
//  Pix2Pix or the ResNeXt-512 step, interleaved same-box runs)
static int ew_grid(int64_t nvec) {
    int64_t b = (nvec + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}

extern "C" int pai_bn_apply(int dtype, const void* z, int64_t M, int C, const float* scale,
                            const float* shift, int act, void* out, void* stream) {
    PAI_CHECK(z && out && scale && shift, "pai_bn_apply: null pointer");
    PAI_CHECK(C % 8 == 0, "pai_bn_apply: C=%d must be a multiple of 8", C);
    const int64_t nvec = M * C / 8;
    hipStream_t s = (hipStream_t)stream;
    if (const int r = ew_stream_bn_apply(dtype, z, M, C, scale, shift, act, out, s); r >= 0) return r;
    if (dtype == PAI_F32)
        PAI_LAUNCH(bn_apply_k<float>, dim3(ew_grid(nvec)), dim3(256), 0, s, (const float*)z, nvec, C,
                           scale, shift, act, (float*)out);
    else
        PAI_LAUNCH(bn_apply_k<bf16_t>, dim3(ew_grid(nvec)), dim3(256), 0, s, (const bf16_t*)z, nvec,
                           C, scale, shift, act, (bf16_t*)out);
    PAI_LAUNCH_CHECK();
    return 0;
}

// out = act(act_a(za * sca + sha) + (zb * scb + shb)): the tail of a residual block -- BatchNorm (+ ReLU: the ResNeXt block,
// reference models/res_unet.py:160-163) of the residual branch, BatchNorm of the skip branch (scb == NULL: the skip is the
// identity, zb is added as it is), the sum and the ReLU behind it (models/res_unet.py:74,105,165-171,
// models/trans_unet.py:227-236) in ONE pass over three tensors instead of bn_apply + bn_apply + add_act over seven.
// fp32: the same roundings as the three passes (fma, fma, add).
template <typename T>
__global__ __launch_bounds__(256) void bn2_add_act_k(const T* za, const float* sca, const float* sha, const T* zb,
                                                     const float* scb, const float* shb, int64_t nvec, int C, int act_a,
                                                     int act, T* out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        const int c0 = (int)((i * 8) % C);
        float a[8], b[8], s1[8], h1[8];
        V8<T>::ld(za + i * 8, a);
        V8<T>::ld(zb + i * 8, b);
        V8<float>::ld(sca + c0, s1);
        V8<float>::ld(sha + c0, h1);
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = act_apply(fmaf(a[k], s1[k], h1[k]), act_a);
        if (scb) {
            V8<float>::ld(scb + c0, s1);
            V8<float>::ld(shb + c0, h1);
#pragma unroll
            for (int k = 0; k < 8; ++k) b[k] = fmaf(b[k], s1[k], h1[k]);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = act_apply(a[k] + b[k], act);
        V8<T>::st(out + i * 8, a);
    }
}

extern "C" int pai_bn2_add_act(int dtype, const void* za, const float* scale_a, const float* shift_a, const void* zb,
                               const float* scale_b, const float* shift_b, int64_t M, int C, int act_a, int act, void* out,
                               void* stream) {
    PAI_CHECK(za && zb && out && scale_a && shift_a, "pai_bn2_add_act: null pointer");
    PAI_CHECK((scale_b == nullptr) == (shift_b == nullptr), "pai_bn2_add_act: scale_b and shift_b go together");
    PAI_CHECK(C % 8 == 0, "pai_bn2_add_act: C=%d must be a multiple of 8", C);
    for (int t : {act_a, act})
        PAI_CHECK(t == PAI_ACT_NONE || t == PAI_ACT_RELU || t == PAI_ACT_LRELU, "pai_bn2_add_act: act=%d", t);
    const int64_t nvec = M * C / 8;
    hipStream_t s = (hipStream_t)stream;
    if (const int r = ew_stream_bn2_add_act(dtype, za, scale_a, shift_a, zb, scale_b, shift_b, M, C, act_a, act, out, s); r >= 0)
        return r;
    if (dtype == PAI_F32)
        PAI_LAUNCH(bn2_add_act_k<float>, dim3(ew_grid(nvec)), dim3(256), 0, s, (const float*)za, scale_a, shift_a,
                   (const float*)zb, scale_b, shift_b, nvec, C, act_a, act, (float*)out);
    else
        PAI_LAUNCH(bn2_add_act_k<bf16_t>, dim3(ew_grid(nvec)), dim3(256), 0, s, (const bf16_t*)za, scale_a, shift_a,
                   (const bf16_t*)zb, scale_b, shift_b, nvec, C, act_a, act, (bf16_t*)out);
    PAI_LAUNCH_CHECK();
    return 0;
}

// ---- backward -------------------------------------------------------------------------------
constexpr int BWD_MAX_PARTIAL = 2048;

extern "C" int pai_bn_bwd_partial_rows(int64_t M) {
    int64_t r = (M + 63) / 64;
    if (r > BWD_MAX_PARTIAL) r = BWD_MAX_PARTIAL;
    if (r < 1) r = 1;
    return (int)r;
}

// thread = (8-channel group, row lane); block = one contiguous slab of rows
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_k(const T* g1, int act1, const T* g2, int act2,
                                                       const T* a, const T* z, int64_t M, int C,
                                                       int64_t rows_per_block, const float* mean,
                                                       const float* rstd, T* du, float* partials,
                                                       const float* scale = nullptr, const float* shift = nullptr) {
    __shared__ float red[2][256][8];
    const int groups = C / 8;
    const int tid = threadIdx.x;
    float s1[8], s2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s1[k] = s2[k] = 0.f;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(M, r0 + rows_per_block);
    // C/8 may exceed 256 (C = 4096 not expected) -> loop over channel groups in strides
    for (int cg0 = 0; cg0 < groups; cg0 += 256) {
        const int per_pass = min(groups - cg0, 256);
        const int lanes = 256 / per_pass;  // per_pass is a power of two <= 256
        const int cg = cg0 + tid % per_pass, rl = tid / per_pass;
        float mu[8], rs[8], sc[8], sh[8];
        V8<float>::ld(mean + cg * 8, mu);
        V8<float>::ld(rstd + cg * 8, rs);
        if (scale) { V8<float>::ld(scale + cg * 8, sc); V8<float>::ld(shift + cg * 8, sh); }
#pragma unroll
        for (int k = 0; k < 8; ++k) s1[k] = s2[k] = 0.f;
        if (rl < lanes) {
            struct Row { float gv[8], zv[8], av[8], g2v[8]; };
            auto load = [&](int64_t r, Row& q) {
                const int64_t off = r * C + cg * 8;
                V8<T>::ld(g1 + off, q.gv);
                V8<T>::ld(z + off, q.zv);
                if (a) V8<T>::ld(a + off, q.av);
                if (g2) V8<T>::ld(g2 + off, q.g2v);
                if (scale) {   // sign source: the pre-activation rebuilt from z (pai_conv_dgrad_bn's two-pass form)
#pragma unroll
                    for (int k = 0; k < 8; ++k) q.av[k] = fmaf(q.zv[k], sc[k], sh[k]);
                }
            };
            auto consume = [&](int64_t r, const Row& q) {
                float d[8];
                if (a || scale) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) d[k] = q.gv[k] * act_grad(q.av[k], act1);
                    if (g2) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) d[k] = fmaf(q.g2v[k], act_grad(q.av[k], act2), d[k]);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) d[k] = q.gv[k];
                    if (g2) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) d[k] += q.g2v[k];
                    }
                }
                if (du) V8<T>::st(du + r * C + cg * 8, d);     // du == null: du IS g1 (no activation, no second gradient)
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    // statistics from the value as stored (what pass 2 will read back)
                    float dd = d[k];
                    if (sizeof(T) == 2) dd = bf2f(f2bf(dd));
                    s1[k] += dd;
                    s2[k] = fmaf(dd, (q.zv[k] - mu[k]) * rs[k], s2[k]);
                }
            };
            // two rows in flight per thread; row order (and so the summation order) is unchanged
            int64_t r = r0 + rl;
            for (; r + lanes < r1; r += 2 * lanes) {
                Row q0, q1;
                load(r, q0);
                load(r + lanes, q1);
                consume(r, q0);
                consume(r + lanes, q1);
            }
            if (r < r1) {
                Row q0;
                load(r, q0);
                consume(r, q0);
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) { red[0][tid][k] = s1[k]; red[1][tid][k] = s2[k]; }
        __syncthreads();
        if (tid < per_pass) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float t1 = 0.f, t2 = 0.f;
                for (int l = 0; l < lanes; ++l) { t1 += red[0][tid + l * per_pass][k]; t2 += red[1][tid + l * per_pass][k]; }
                // partial row layout: [block][0]=sum(du) -> dbeta, [block][1]=sum(du*xhat) -> dgamma
                partials[((size_t)blockIdx.x * 2 + 0) * C + cg * 8 + k] = t1;
                partials[((size_t)blockIdx.x * 2 + 1) * C + cg * 8 + k] = t2;
            }
        }
        __syncthreads();
    }
}

// 8 channels x FIN_LANES row lanes per block: the partial rows (up to 4096 from the fused input-gradient stores)
// are summed in fp64; the launch is a chain of dependent L2 round trips, so it is as wide as a block can be
// (measured and dropped: 8 rows in flight per thread -- 20.4 -> 25.5 us per launch)
constexpr int FIN_LANES = 128;
__global__ __launch_bounds__(8 * FIN_LANES) void bn_bwd_finalize_k(const float* partials, int rows, int C, float* sums,
                                                                   float* dgamma, float* dbeta) {
    __shared__ double red[2][FIN_LANES][8];
    const int cl = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int c = blockIdx.x * 8 + cl;
    double s1 = 0.0, s2 = 0.0;
    if (c < C) {
        // (8 rows in flight per lane were tried twice: faster alone, 11 -> 6 us at 4096 rows, but 7.4 -> 22.9 us on average
        //  in the training step, where the launch runs beside a weight-gradient kernel -- rocprofv3 traces of round 3)
        int r = rl;
        for (; r < rows; r += FIN_LANES) {
            s1 += (double)partials[((size_t)r * 2 + 0) * C + c];
            s2 += (double)partials[((size_t)r * 2 + 1) * C + c];
        }
    }
    // the 8 row lanes of a wave meet by shuffles, the 16 waves through LDS (the 128-term serial sum from LDS that
    // stood here was most of the launch)
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {
        s1 += __shfl_xor(s1, o, 64);
        s2 += __shfl_xor(s2, o, 64);
    }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) < 8) {
        red[0][wv][cl] = s1;
        red[1][wv][cl] = s2;
    }
    __syncthreads();
    if (rl == 0 && c < C) {
        s1 = s2 = 0.0;
#pragma unroll
        for (int l = 0; l < FIN_LANES / 8; ++l) { s1 += red[0][l][cl]; s2 += red[1][l][cl]; }
        sums[c] = (float)s1;
        sums[C + c] = (float)s2;
        if (dbeta) dbeta[c] += (float)s1;
        if (dgamma) dgamma[c] += (float)s2;
    }
}

// pai_conv_dgrad_bn for the kernel families without the fused store: in place over the plain input gradient
int bn_bwd_reduce_affine(int dtype, void* g1_du, int act1, const void* g2, int act2, const void* z, int64_t M, int C,
                         const float* scale, const float* shift, const float* mean, const float* rstd,
                         float* partials, hipStream_t s) {
    PAI_CHECK(C % 8 == 0 && ((C / 8) & (C / 8 - 1)) == 0, "pai_conv_dgrad_bn: C1=%d must be 8 * 2^k", C);
    const int rows = pai_bn_bwd_partial_rows(M);
    const int64_t rpb = (M + rows - 1) / rows;
    // without an affine map the stored tensor itself carries the sign
    const void* a = scale ? nullptr : z;
    if (dtype == PAI_F32)
        PAI_LAUNCH(bn_bwd_reduce_k<float>, dim3(rows), dim3(256), 0, s, (const float*)g1_du, act1,
                           (const float*)g2, act2, (const float*)a, (const float*)z, M, C, rpb, mean, rstd,
                           (float*)g1_du, partials, scale, shift);
    else
        PAI_LAUNCH(bn_bwd_reduce_k<bf16_t>, dim3(rows), dim3(256), 0, s, (const bf16_t*)g1_du, act1,
                           (const bf16_t*)g2, act2, (const bf16_t*)a, (const bf16_t*)z, M, C, rpb, mean, rstd,
                           (bf16_t*)g1_du, partials, scale, shift);
    PAI_LAUNCH_CHECK();
    return 0;
}

extern "C" int pai_bn_bwd_finalize(const float* partials, int rows, int C, float* sums, float* dgamma,
                                   float* dbeta, void* stream) {
    PAI_CHECK(partials && sums && rows > 0 && C > 0, "pai_bn_bwd_finalize: bad arguments");
    PAI_LAUNCH(bn_bwd_finalize_k, dim3(cdiv(C, 8)), dim3(8 * FIN_LANES), 0, (hipStream_t)stream, partials, rows, C,
                       sums, dgamma, dbeta);
    PAI_LAUNCH_CHECK();
    return 0;
}

extern "C" int pai_bn_bwd_reduce(int dtype, const void* g1, int act1, const void* g2, int act2,
                                 const void* a, const void* z, int64_t M, int C, const float* mean,
                                 const float* rstd, void* du, float* partials, float* sums,
                                 float* dgamma, float* dbeta, void* stream) {
    PAI_CHECK(g1 && z && partials && sums && mean && rstd, "pai_bn_bwd_reduce: null pointer");
    PAI_CHECK(du || (!g2 && !a && act1 == PAI_ACT_NONE), "pai_bn_bwd_reduce: du may be null only when du == g1");
    PAI_CHECK(C % 8 == 0 && ((C / 8) & (C / 8 - 1)) == 0, "pai_bn_bwd_reduce: C=%d must be 8 * 2^k", C);
    PAI_CHECK(a || (act1 == PAI_ACT_NONE && act2 == PAI_ACT_NONE), "pai_bn_bwd_reduce: act without a");
    hipStream_t s = (hipStream_t)stream;
    const int rows = pai_bn_bwd_partial_rows(M);
    const int64_t rpb = (M + rows - 1) / rows;
    int taken = -1;
    if (!g2 && !a && !du) taken = ew_stream_bn_bwd_reduce(dtype, g1, PAI_ACT_NONE, z, M, C, nullptr, nullptr, mean, rstd, partials, rows, s);
    if (taken > 0) return taken;
    if (taken == 0) {
    } else if (dtype == PAI_F32)
        PAI_LAUNCH(bn_bwd_reduce_k<float>, dim3(rows), dim3(256), 0, s, (const float*)g1, act1,
                           (const float*)g2, act2, (const float*)a, (const float*)z, M, C, rpb, mean, rstd,
                           (float*)du, partials, nullptr, nullptr);
    else
        PAI_LAUNCH(bn_bwd_reduce_k<bf16_t>, dim3(rows), dim3(256), 0, s, (const bf16_t*)g1, act1,
                           (const bf16_t*)g2, act2, (const bf16_t*)a, (const bf16_t*)z, M, C, rpb, mean, rstd,
                           (bf16_t*)du, partials, nullptr, nullptr);
    PAI_LAUNCH_CHECK();
    PAI_LAUNCH(bn_bwd_finalize_k, dim3(cdiv(C, 8)), dim3(8 * FIN_LANES), 0, s, partials, rows, C, sums, dgamma,
                       dbeta);
    PAI_LAUNCH_CHECK();
    return 0;
}

// The same pass with the activation's sign rebuilt from z (pre = z * scale + shift) instead of read from the stored
// activated output: one tensor read less (6 instead of 7 tensor passes per BatchNorm backward of the composable
// networks, whose residual U-Net step is bound by exactly these passes).
extern "C" int pai_bn_bwd_reduce_affine(int dtype, const void* g1, int act1, const void* g2, int act2,
                                        const void* z, int64_t M, int C, const float* scale, const float* shift,
                                        const float* mean, const float* rstd, void* du, float* partials,
                                        float* sums, float* dgamma, float* dbeta, void* stream) {
    PAI_CHECK(g1 && z && partials && sums && mean && rstd && scale && shift, "pai_bn_bwd_reduce_affine: null pointer");
    PAI_CHECK(du || !g2, "pai_bn_bwd_reduce_affine: du may be null only without a second gradient (pass 2 rebuilds du from g1)");
    PAI_CHECK(C % 8 == 0 && ((C / 8) & (C / 8 - 1)) == 0, "pai_bn_bwd_reduce_affine: C=%d must be 8 * 2^k", C);
    hipStream_t s = (hipStream_t)stream;
    const int rows = pai_bn_bwd_partial_rows(M);
    const int64_t rpb = (M + rows - 1) / rows;
    int taken = -1;
    if (!g2 && !du) taken = ew_stream_bn_bwd_reduce(dtype, g1, act1, z, M, C, scale, shift, mean, rstd, partials, rows, s);
    if (taken > 0) return taken;
    if (taken == 0) {
    } else if (dtype == PAI_F32)
        PAI_LAUNCH(bn_bwd_reduce_k<float>, dim3(rows), dim3(256), 0, s, (const float*)g1, act1,
                           (const float*)g2, act2, (const float*)nullptr, (const float*)z, M, C, rpb, mean, rstd,
                           (float*)du, partials, scale, shift);
    else
        PAI_LAUNCH(bn_bwd_reduce_k<bf16_t>, dim3(rows), dim3(256), 0, s, (const bf16_t*)g1, act1,
                           (const bf16_t*)g2, act2, (const bf16_t*)nullptr, (const bf16_t*)z, M, C, rpb, mean, rstd,
                           (bf16_t*)du, partials, scale, shift);
    PAI_LAUNCH_CHECK();
    PAI_LAUNCH(bn_bwd_finalize_k, dim3(cdiv(C, 8)), dim3(8 * FIN_LANES), 0, s, partials, rows, C, sums, dgamma,
                       dbeta);
    PAI_LAUNCH_CHECK();
    return 0;
}

// act1 >= 0 (pai_bn_bwd_apply_affine): `du` is the gradient behind the activation and du proper is rebuilt here as pass 1
// formed it -- g * act1'(z * scale + shift), rounded to the storage type -- so pass 1 need not store it
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_k(const T* du, const T* z, int64_t nvec, int C,
                                                      float inv_m, const float* mean, const float* rstd,
                                                      const float* gamma, const float* sums, T* dz, int act1 = -1,
                                                      const float* scale = nullptr, const float* shift = nullptr) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        const int c0 = (int)((i * 8) % C);
        float d[8], zv[8], mu[8], rs[8], gm[8], sb[8], sg[8];
        V8<T>::ld(du + i * 8, d);
        V8<T>::ld(z + i * 8, zv);
        if (act1 >= 0) {
            float sc[8], sh[8];
            V8<float>::ld(scale + c0, sc);
            V8<float>::ld(shift + c0, sh);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                d[k] = d[k] * act_grad(fmaf(zv[k], sc[k], sh[k]), act1);
                if (sizeof(T) == 2) d[k] = bf2f(f2bf(d[k]));
            }
        }
        V8<float>::ld(mean + c0, mu);
        V8<float>::ld(rstd + c0, rs);
        V8<float>::ld(sums + c0, sb);
        V8<float>::ld(sums + C + c0, sg);
        if (gamma) V8<float>::ld(gamma + c0, gm);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float xh = (zv[k] - mu[k]) * rs[k];
            const float g = gamma ? gm[k] : 1.f;
            d[k] = g * rs[k] * (d[k] - sb[k] * inv_m - xh * sg[k] * inv_m);
        }
        V8<T>::st(dz + i * 8, d);
    }
}

extern "C" int pai_bn_bwd_apply(int dtype, const void* du, const void* z, int64_t M, int C,
                                const float* mean, const float* rstd, const float* gamma,
                                const float* sums, void* dz, void* stream) {
    PAI_CHECK(du && z && dz && mean && rstd && sums, "pai_bn_bwd_apply: null pointer");
    PAI_CHECK(C % 8 == 0, "pai_bn_bwd_apply: C=%d must be a multiple of 8", C);
    const int64_t nvec = M * C / 8;
    const float inv_m = (float)(1.0 / (double)M);
    hipStream_t s = (hipStream_t)stream;
    if (const int r = ew_stream_bn_bwd_apply(dtype, du, PAI_ACT_NONE, z, M, C, nullptr, nullptr, mean, rstd, gamma, sums, dz, s); r >= 0)
        return r;
    if (dtype == PAI_F32)
        PAI_LAUNCH(bn_bwd_apply_k<float>, dim3(ew_grid(nvec)), dim3(256), 0, s, (const float*)du,
                           (const float*)z, nvec, C, inv_m, mean, rstd, gamma, sums, (float*)dz, -1, nullptr, nullptr);
    else
        PAI_LAUNCH(bn_bwd_apply_k<bf16_t>, dim3(ew_grid(nvec)), dim3(256), 0, s, (const bf16_t*)du,
                           (const bf16_t*)z, nvec, C, inv_m, mean, rstd, gamma, sums, (bf16_t*)dz, -1, nullptr, nullptr);
    PAI_LAUNCH_CHECK();
    return 0;
}

// Pass 2 for a pass 1 that did not store du (pai_bn_bwd_reduce_affine with du = NULL): g1 is the gradient behind the
// activation, du = g1 * act1'(z * scale + shift) is rebuilt on the fly -- 5 tensor passes per BatchNorm backward instead of 6.
extern "C" int pai_bn_bwd_apply_affine(int dtype, const void* g1, int act1, const void* z, int64_t M, int C,
                                       const float* scale, const float* shift, const float* mean, const float* rstd,
                                       const float* gamma, const float* sums, void* dz, void* stream) {
    PAI_CHECK(g1 && z && dz && mean && rstd && sums && scale && shift, "pai_bn_bwd_apply_affine: null pointer");
    PAI_CHECK(C % 8 == 0, "pai_bn_bwd_apply_affine: C=%d must be a multiple of 8", C);
    PAI_CHECK(act1 >= 0 && act1 <= PAI_ACT_TANH, "pai_bn_bwd_apply_affine: bad activation %d", act1);
    const int64_t nvec = M * C / 8;
    const float inv_m = (float)(1.0 / (double)M);
    hipStream_t s = (hipStream_t)stream;
    if (const int r = ew_stream_bn_bwd_apply(dtype, g1, act1, z, M, C, scale, shift, mean, rstd, gamma, sums, dz, s); r >= 0) return r;
    if (dtype == PAI_F32)
        PAI_LAUNCH(bn_bwd_apply_k<float>, dim3(ew_grid(nvec)), dim3(256), 0, s, (const float*)g1,
                           (const float*)z, nvec, C, inv_m, mean, rstd, gamma, sums, (float*)dz, act1, scale, shift);
    else
        PAI_LAUNCH(bn_bwd_apply_k<bf16_t>, dim3(ew_grid(nvec)), dim3(256), 0, s, (const bf16_t*)g1,
                           (const bf16_t*)z, nvec, C, inv_m, mean, rstd, gamma, sums, (bf16_t*)dz, act1, scale, shift);
    PAI_LAUNCH_CHECK();
    return 0;
}

// ---- the tail of a residual block, backward (reference models/res_unet.py:165-171: conv_block(x) + conv_skip(x), both ending
// in a BatchNorm): the residual branch's BatchNorm (with its ReLU, act_a) and the skip branch's read the SAME gradient d.
// pai_bn2_bwd_reduce = pai_bn_bwd_reduce(_affine) for both (du not stored) + the two finalizes; pai_bn2_bwd_apply = both
// pai_bn_bwd_apply(_affine).  Big bf16 tensors: one pass over (d, za, zb) each (3 + 5 tensor passes instead of 4 + 6);
// everything else: the one-branch calls, twice.
extern "C" int pai_bn2_bwd_reduce(int dtype, const void* d, int act_a, const void* za, const void* zb, int64_t M, int C,
                                  const float* scale_a, const float* shift_a, const float* mean_a, const float* rstd_a,
                                  const float* mean_b, const float* rstd_b, float* part_a, float* part_b, float* sums_a,
                                  float* sums_b, void* stream) {
    PAI_CHECK(d && za && zb && mean_a && rstd_a && mean_b && rstd_b && part_a && part_b && sums_a && sums_b,
              "pai_bn2_bwd_reduce: null pointer");
    PAI_CHECK(C % 8 == 0 && ((C / 8) & (C / 8 - 1)) == 0, "pai_bn2_bwd_reduce: C=%d must be 8 * 2^k", C);
    PAI_CHECK(act_a == PAI_ACT_NONE || (scale_a && shift_a), "pai_bn2_bwd_reduce: an activation needs scale_a / shift_a");
    hipStream_t s = (hipStream_t)stream;
    const int rows = pai_bn_bwd_partial_rows(M);
    const int r = ew_stream_bn2_bwd_reduce(dtype, d, act_a, za, zb, M, C, scale_a, shift_a, mean_a, rstd_a, mean_b, rstd_b, part_a,
                                           part_b, rows, s);
    if (r > 0) return r;
    if (r == 0) {
        if (int rc = pai_bn_bwd_finalize(part_a, rows, C, sums_a, nullptr, nullptr, stream)) return rc;
        return pai_bn_bwd_finalize(part_b, rows, C, sums_b, nullptr, nullptr, stream);
    }
    int rc = act_a != PAI_ACT_NONE
                 ? pai_bn_bwd_reduce_affine(dtype, d, act_a, nullptr, PAI_ACT_NONE, za, M, C, scale_a, shift_a, mean_a, rstd_a,
                                            nullptr, part_a, sums_a, nullptr, nullptr, stream)
                 : pai_bn_bwd_reduce(dtype, d, PAI_ACT_NONE, nullptr, PAI_ACT_NONE, nullptr, za, M, C, mean_a, rstd_a, nullptr,
                                     part_a, sums_a, nullptr, nullptr, stream);
    if (rc) return rc;
    return pai_bn_bwd_reduce(dtype, d, PAI_ACT_NONE, nullptr, PAI_ACT_NONE, nullptr, zb, M, C, mean_b, rstd_b, nullptr, part_b,
                             sums_b, nullptr, nullptr, stream);
}

extern "C" int pai_bn2_bwd_apply(int dtype, const void* d, int act_a, const void* za, const void* zb, int64_t M, int C,
                                 const float* scale_a, const float* shift_a, const float* mean_a, const float* rstd_a,
                                 const float* gamma_a, const float* sums_a, const float* mean_b, const float* rstd_b,
                                 const float* gamma_b, const float* sums_b, void* dza, void* dzb, void* stream) {
    PAI_CHECK(d && za && zb && dza && dzb && mean_a && rstd_a && mean_b && rstd_b && sums_a && sums_b,
              "pai_bn2_bwd_apply: null pointer");
    PAI_CHECK(C % 8 == 0, "pai_bn2_bwd_apply: C=%d must be a multiple of 8", C);
    PAI_CHECK(act_a == PAI_ACT_NONE || (scale_a && shift_a), "pai_bn2_bwd_apply: an activation needs scale_a / shift_a");
    const int r = ew_stream_bn2_bwd_apply(dtype, d, act_a, za, zb, M, C, scale_a, shift_a, mean_a, rstd_a, gamma_a, sums_a, mean_b,
                                          rstd_b, gamma_b, sums_b, dza, dzb, (hipStream_t)stream);
    if (r >= 0) return r;
    int rc = act_a != PAI_ACT_NONE
                 ? pai_bn_bwd_apply_affine(dtype, d, act_a, za, M, C, scale_a, shift_a, mean_a, rstd_a, gamma_a, sums_a, dza, stream)
                 : pai_bn_bwd_apply(dtype, d, za, M, C, mean_a, rstd_a, gamma_a, sums_a, dza, stream);
    if (rc) return rc;
    return pai_bn_bwd_apply(dtype, d, zb, M, C, mean_b, rstd_b, gamma_b, sums_b, dzb, stream);
}

// ---- finalize + apply in ONE launch for the small layers (the U-Net bottleneck: <= 4096 rows, <= 256 partial rows) -------
// pai_bn_finalize -> pai_bn_apply and pai_bn_bwd_finalize -> pai_bn_bwd_apply are two dependent launches of 5-9 us each
// around tensors of 0.1-4 MB: twelve such pairs per Pix2Pix step (encoders[4-6], decoders[0-2], forward and backward).
// Here every workgroup (64 channels x a block of rows) first re-derives the per-channel totals of its 64 channels from the
// partial rows -- <= 128 KB from L2, summed in EXACTLY the order of bn_finalize_wide_k / bn_finalize_k / bn_bwd_finalize_k
// (128 row lanes, 8-lane butterflies, 16 waves in order; <= 16 rows: one sequential sum), so the statistics, and with
// them every output, are bit-identical to the two-launch form -- and then applies them to its rows.  The workgroups of
// row block 0 write the per-channel outputs and advance the running statistics.
constexpr int FA_ROWS = 128;      // rows of the tensor per workgroup

// totals of channel c over the partial rows [R][2][C], component `comp`, in the finalize kernels' summation order
__device__ __forceinline__ void fa_wave_partials(const float* part, int R, int C, int c, int w, double (*Wl)[16][64]) {
    // thread (c, w): "wave" w of the finalize kernel = the butterfly of its 8 row lanes
    const int cl = threadIdx.x & 63;
    {
        double P[2][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int rl = 8 * w + j;
            double a = 0.0, b = 0.0;
            for (int r = rl; r < R; r += 128) {
                a += (double)part[((size_t)r * 2 + 0) * C + c];
                b += (double)part[((size_t)r * 2 + 1) * C + c];
            }
            P[0][j] = a;
            P[1][j] = b;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k)
            Wl[k][w][cl] = ((P[k][0] + P[k][1]) + (P[k][2] + P[k][3])) + ((P[k][4] + P[k][5]) + (P[k][6] + P[k][7]));
    }
}
__device__ __forceinline__ void fa_totals(const float* part, int R, int C, int c, bool wide, double (*Wl)[16][64], double& s, double& q) {
    const int cl = threadIdx.x & 63;
    s = q = 0.0;
    if (wide) {
#pragma unroll
        for (int l = 0; l < 16; ++l) { s += Wl[0][l][cl]; q += Wl[1][l][cl]; }
    } else {
        for (int r = 0; r < R; ++r) {
            s += (double)part[((size_t)r * 2 + 0) * C + c];
            q += (double)part[((size_t)r * 2 + 1) * C + c];
        }
    }
}

template <typename T>
__global__ __launch_bounds__(1024) void bn_fin_apply_k(const float* stats, int R, int wide, int C, double count, const float* gamma,
                                                      const float* beta, float eps, float momentum, int n_updates,
                                                      float* running_mean, float* running_var, int64_t* nbt, float* mean_o,
                                                      float* rstd_o, float* scale_o, float* shift_o, const T* z, int64_t M,
                                                      int act, T* out) {
    __shared__ double Wl[2][16][64];
    __shared__ float coef[2][64];
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    if (wide) fa_wave_partials(stats, R, C, c, g, Wl);
    __syncthreads();
    if (g == 0) {
        double s, q;
        fa_totals(stats, R, C, c, wide != 0, Wl, s, q);
        const double mean = s / count;
        double var = q / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        const float gm = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
        const float sc = gm * rstd;
        const float sh = b - (float)mean * sc;
        coef[0][cl] = sc;
        coef[1][cl] = sh;
        if (blockIdx.y == 0) {
            if (blockIdx.x == 0 && cl == 0 && nbt) *nbt += n_updates;
            mean_o[c] = (float)mean;
            rstd_o[c] = rstd;
            scale_o[c] = sc;
            shift_o[c] = sh;
            if (running_mean && running_var) {
                const float unbiased = (float)(count > 1.0 ? var * count / (count - 1.0) : var);
                float rm = running_mean[c], rv = running_var[c];
                for (int u = 0; u < n_updates; ++u) {
                    rm = (1.f - momentum) * rm + momentum * (float)mean;
                    rv = (1.f - momentum) * rv + momentum * unbiased;
                }
                running_mean[c] = rm;
                running_var[c] = rv;
            }
        }
    }
    __syncthreads();
    // apply: thread = (row lane, 8-channel group) of the workgroup's FA_ROWS x 64 block; same arithmetic as bn_apply_k
    const int cg = threadIdx.x & 7, r0 = threadIdx.x >> 3;
    float sc[8], sh[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { sc[k] = coef[0][cg * 8 + k]; sh[k] = coef[1][cg * 8 + k]; }
    const int64_t m0 = (int64_t)blockIdx.y * FA_ROWS;
    for (int r = r0; r < FA_ROWS; r += 128) {
        const int64_t m = m0 + r;
        if (m >= M) break;
        const size_t off = (size_t)m * C + blockIdx.x * 64 + cg * 8;
        float v[8];
        V8<T>::ld(z + off, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = act_apply(fmaf(v[k], sc[k], sh[k]), act);
        V8<T>::st(out + off, v);
    }
}

template <typename T>
__global__ __launch_bounds__(1024) void bn_bwd_fin_apply_k(const float* partials, int R, int C, float* sums, float* dgamma,
                                                          float* dbeta, const T* du, const T* z, int64_t M, float inv_m,
                                                          const float* mean, const float* rstd, const float* gamma, T* dz) {
    __shared__ double Wl[2][16][64];
    __shared__ float coef[2][64];
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    fa_wave_partials(partials, R, C, c, g, Wl);
    __syncthreads();
    if (g == 0) {
        double s1, s2;
        fa_totals(partials, R, C, c, true, Wl, s1, s2);
        coef[0][cl] = (float)s1;
        coef[1][cl] = (float)s2;
        if (blockIdx.y == 0) {
            sums[c] = (float)s1;
            sums[C + c] = (float)s2;
            if (dbeta) dbeta[c] += (float)s1;
            if (dgamma) dgamma[c] += (float)s2;
        }
    }
    __syncthreads();
    const int cg = threadIdx.x & 7, r0 = threadIdx.x >> 3;
    const int cb = blockIdx.x * 64 + cg * 8;
    float mu[8], rs[8], gm[8], sb[8], sg[8];
    V8<float>::ld(mean + cb, mu);
    V8<float>::ld(rstd + cb, rs);
    if (gamma) V8<float>::ld(gamma + cb, gm);
#pragma unroll
    for (int k = 0; k < 8; ++k) { sb[k] = coef[0][cg * 8 + k]; sg[k] = coef[1][cg * 8 + k]; }
    const int64_t m0 = (int64_t)blockIdx.y * FA_ROWS;
    for (int r = r0; r < FA_ROWS; r += 128) {
        const int64_t m = m0 + r;
        if (m >= M) break;
        const size_t off = (size_t)m * C + cb;
        float d[8], zv[8];
        V8<T>::ld(du + off, d);
        V8<T>::ld(z + off, zv);
#pragma unroll
        for (int k = 0; k < 8; ++k) {      // (same expression as bn_bwd_apply_k)
            const float xh = (zv[k] - mu[k]) * rs[k];
            const float gk = gamma ? gm[k] : 1.f;
            d[k] = gk * rs[k] * (d[k] - sb[k] * inv_m - xh * sg[k] * inv_m);
        }
        V8<T>::st(dz + off, d);
    }
}

// the layers the fused launches take: channel groups of 64, a small tensor, few partial rows (tunable bn_fuse_small)
bool bn_fuse_small_ok(int rows, int64_t M, int C) {
    return pai_tunable("bn_fuse_small", 1) && (C % 64) == 0 && M > 0 && M <= 4096 && rows > 0 && rows <= 256;
}

int launch_bn_fin_apply(int dtype, const float* stats, int rows, int C, int64_t count, const float* gamma, const float* beta,
                        float eps, float momentum, int n_updates, float* running_mean, float* running_var, int64_t* nbt,
                        float* mean, float* rstd, float* scale, float* shift, const void* z, int act, void* out,
                        hipStream_t s) {
    const dim3 grid(C / 64, cdiv(count, FA_ROWS));
    const int wide = rows > 16;      // pai_bn_finalize: more than 16 partial rows go through bn_finalize_wide_k's tree
    if (dtype == PAI_F32)
        PAI_LAUNCH(bn_fin_apply_k<float>, grid, dim3(1024), 0, s, stats, rows, wide, C, (double)count, gamma, beta, eps, momentum,
                   n_updates, running_mean, running_var, nbt, mean, rstd, scale, shift, (const float*)z, count, act, (float*)out);
    else
        PAI_LAUNCH(bn_fin_apply_k<bf16_t>, grid, dim3(1024), 0, s, stats, rows, wide, C, (double)count, gamma, beta, eps, momentum,
                   n_updates, running_mean, running_var, nbt, mean, rstd, scale, shift, (const bf16_t*)z, count, act, (bf16_t*)out);
    PAI_LAUNCH_CHECK();
    return 0;
}

int launch_bn_bwd_fin_apply(int dtype, const float* partials, int rows, int C, float* sums, float* dgamma, float* dbeta,
                            const void* du, const void* z, int64_t M, const float* mean, const float* rstd,
                            const float* gamma, void* dz, hipStream_t s) {
    const dim3 grid(C / 64, cdiv(M, FA_ROWS));
    const float inv_m = (float)(1.0 / (double)M);
    if (dtype == PAI_F32)
        PAI_LAUNCH(bn_bwd_fin_apply_k<float>, grid, dim3(1024), 0, s, partials, rows, C, sums, dgamma, dbeta, (const float*)du,
                   (const float*)z, M, inv_m, mean, rstd, gamma, (float*)dz);
    else
        PAI_LAUNCH(bn_bwd_fin_apply_k<bf16_t>, grid, dim3(1024), 0, s, partials, rows, C, sums, dgamma, dbeta, (const bf16_t*)du,
                   (const bf16_t*)z, M, inv_m, mean, rstd, gamma, (bf16_t*)dz);
    PAI_LAUNCH_CHECK();
    return 0;
}

// ---- activation backward without a norm ------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void act_bwd_k(const T* g1, int act1, const T* g2, int act2, const T* a,
                                                 int64_t nvec, T* du) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        float gv[8], av[8], d[8];
        V8<T>::ld(g1 + i * 8, gv);
        V8<T>::ld(a + i * 8, av);
#pragma unroll
        for (int k = 0; k < 8; ++k) d[k] = gv[k] * act_grad(av[k], act1);
        if (g2) {
            float g2v[8];
            V8<T>::ld(g2 + i * 8, g2v);
#pragma unroll
            for (int k = 0; k < 8; ++k) d[k] = fmaf(g2v[k], act_grad(av[k], act2), d[k]);
        }
        V8<T>::st(du + i * 8, d);
    }
}

extern "C" int pai_act_bwd(int dtype, const void* g1, int act1, const void* g2, int act2, const void* a,
                           int64_t numel, void* du, void* stream) {
    PAI_CHECK(g1 && a && du, "pai_act_bwd: null pointer");
    PAI_CHECK(numel % 8 == 0, "pai_act_bwd: numel must be a multiple of 8");
    const int64_t nvec = numel / 8;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PAI_F32)
        PAI_LAUNCH(act_bwd_k<float>, dim3(ew_grid(nvec)), dim3(256), 0, s, (const float*)g1, act1,
                           (const float*)g2, act2, (const float*)a, nvec, (float*)du);
    else
        PAI_LAUNCH(act_bwd_k<bf16_t>, dim3(ew_grid(nvec)), dim3(256), 0, s, (const bf16_t*)g1, act1,
                           (const bf16_t*)g2, act2, (const bf16_t*)a, nvec, (bf16_t*)du);
    PAI_LAUNCH_CHECK();
    return 0;
}

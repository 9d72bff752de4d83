// TransUNet pieces that are not convolutions (reference models/trans_unet.py): LayerNorm, exact GELU, the
// multi-head attention core over [S][B][E] tokens, the even-pixel subsample that turns "same" convolutions into
// stride-2 ones, and BatchNorm partial statistics of a tensor no convolution epilogue produced.
// All of it is HBM- or latency-bound row work: one workgroup per token row / attention row, wave reductions,
// 16-B accesses where the channel count allows.
#include "common.h"

__device__ __forceinline__ float block_sum(float v, float* red) {
    // red: >= 8 floats of LDS; every thread of the (<= 512-thread) block calls this
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---------------------------------------------------------------------------------------------------------
// LayerNorm over the last dimension of [M][D] (nn.LayerNorm, models/trans_unet.py:142,144, and norm1 / norm2 of
// nn.TransformerEncoderLayer), with the residual sum in front of it and a broadcast addend behind it fused.
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void layernorm_fwd_k(const T* x, const T* res, int D, const float* gamma,
                                                       const float* beta, float eps, const float* post, int P,
                                                       T* sum_out, T* y, float* mean, float* rstd) {
    extern __shared__ float row[];   // D floats + 8
    float* red = row + D;
    const int64_t m = blockIdx.x;
    const T* xr = x + m * D;
    float s = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) {
        float v = Conv<T>::ld(xr + d);
        if (res) {
            v += Conv<T>::ld(res + m * D + d);
            Conv<T>::st(sum_out + m * D + d, v);
            v = Conv<T>::ld(sum_out + m * D + d);   // normalise the value the backward pass will read
        }
        row[d] = v;
        s += v;
    }
    const float mu = block_sum(s, red) / (float)D;
    float q = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) {
        const float c = row[d] - mu;
        q = fmaf(c, c, q);
    }
    const float var = block_sum(q, red) / (float)D;
    const float r = 1.0f / sqrtf(var + eps);
    if (threadIdx.x == 0) {
        mean[m] = mu;
        rstd[m] = r;
    }
    const float* pr = post ? post + (int64_t)(m % P) * D : nullptr;
    for (int d = threadIdx.x; d < D; d += 256) {
        float v = (row[d] - mu) * r * gamma[d] + beta[d];
        if (pr) v += pr[d];
        Conv<T>::st(y + m * D + d, v);
    }
}

// dx = rstd * (g - mean_D(g) - xhat * mean_D(g * xhat)),  g = dy * gamma
template <typename T>
__global__ __launch_bounds__(256) void layernorm_bwd_dx_k(const T* dy, const T* xs, int D, const float* gamma,
                                                          const float* mean, const float* rstd, T* dx) {
    extern __shared__ float row[];   // D floats (xhat) + 8; g is recomputed in the second pass (dy, gamma are L1 / L2 hits)
    float* red = row + D;
    const int64_t m = blockIdx.x;
    const float mu = mean[m], r = rstd[m];
    float s1 = 0.f, s2 = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) {
        const float xh = (Conv<T>::ld(xs + m * D + d) - mu) * r;
        const float g = Conv<T>::ld(dy + m * D + d) * gamma[d];
        row[d] = xh;
        s1 += g;
        s2 = fmaf(g, xh, s2);
    }
    const float a = block_sum(s1, red) / (float)D;
    const float b = block_sum(s2, red) / (float)D;
    for (int d = threadIdx.x; d < D; d += 256) {
        const float g = Conv<T>::ld(dy + m * D + d) * gamma[d];
        Conv<T>::st(dx + m * D + d, r * (g - a - row[d] * b));
    }
}

// partial[slab][0][d] = sum_rows dy, partial[slab][1][d] = sum_rows dy * xhat   (thread = column, slab = rows)
template <typename T>
__global__ __launch_bounds__(256) void layernorm_bwd_param_k(const T* dy, const T* xs, int64_t M, int D,
                                                             const float* mean, const float* rstd,
                                                             int64_t rows_per_slab, float* partial) {
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d >= D) return;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_slab, r1 = min(M, r0 + rows_per_slab);
    float sb = 0.f, sg = 0.f;
    for (int64_t m = r0; m < r1; ++m) {
        const float g = Conv<T>::ld(dy + m * D + d);
        const float xh = (Conv<T>::ld(xs + m * D + d) - mean[m]) * rstd[m];
        sb += g;
        sg = fmaf(g, xh, sg);
    }
    float* p = partial + (int64_t)blockIdx.y * 2 * D;
    p[d] = sb;
    p[D + d] = sg;
}

extern "C" int pai_layernorm_partial_rows(int64_t M) {
    int64_t slabs = (M + 15) / 16;
    if (slabs > 64) slabs = 64;
    return slabs < 1 ? 1 : (int)slabs;
}

extern "C" int pai_layernorm_fwd(int dtype, const void* x, const void* res, int64_t M, int D, const float* gamma,
                                 const float* beta, float eps, const float* post, int P, void* sum_out, void* y,
                                 float* mean, float* rstd, void* stream) {
    PAI_CHECK(dtype == PAI_F32 || dtype == PAI_BF16, "pai_layernorm_fwd: bad dtype %d", dtype);
    PAI_CHECK(x && gamma && beta && y && mean && rstd && M > 0 && D > 0, "pai_layernorm_fwd: bad arguments");
    PAI_CHECK(D <= 12288, "pai_layernorm_fwd: D=%d exceeds the 12288-element row buffer", D);
    PAI_CHECK(!res || sum_out, "pai_layernorm_fwd: a residual needs sum_out");
    PAI_CHECK(!post || P > 0, "pai_layernorm_fwd: post needs its period P");
    PAI_CHECK(M < ((int64_t)1 << 31), "pai_layernorm_fwd: too many rows");
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = (size_t)(D + 8) * sizeof(float);
    if (dtype == PAI_F32)
        PAI_LAUNCH(layernorm_fwd_k<float>, dim3((unsigned)M), dim3(256), lds, s, (const float*)x,
                           (const float*)res, D, gamma, beta, eps, post, P, (float*)sum_out, (float*)y, mean, rstd);
    else
        PAI_LAUNCH(layernorm_fwd_k<bf16_t>, dim3((unsigned)M), dim3(256), lds, s, (const bf16_t*)x,
                           (const bf16_t*)res, D, gamma, beta, eps, post, P, (bf16_t*)sum_out, (bf16_t*)y, mean, rstd);
    PAI_LAUNCH_CHECK();
    return 0;
}

extern "C" int pai_layernorm_bwd(int dtype, const void* dy, const void* xs, int64_t M, int D, const float* gamma,
                                 const float* mean, const float* rstd, void* dx, float* dgamma_dbeta,
                                 float* partials, void* stream) {
    PAI_CHECK(dtype == PAI_F32 || dtype == PAI_BF16, "pai_layernorm_bwd: bad dtype %d", dtype);
    PAI_CHECK(dy && xs && gamma && mean && rstd && dx && M > 0 && D > 0, "pai_layernorm_bwd: bad arguments");
    PAI_CHECK(D <= 12288, "pai_layernorm_bwd: D=%d exceeds the 12288-element row buffer", D);
    PAI_CHECK(M < ((int64_t)1 << 31), "pai_layernorm_bwd: too many rows");
    PAI_CHECK(!dgamma_dbeta || partials, "pai_layernorm_bwd: parameter gradients need the partials workspace");
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = (size_t)(D + 8) * sizeof(float);
    const int slabs = pai_layernorm_partial_rows(M);
    const int64_t rps = (M + slabs - 1) / slabs;
    if (dtype == PAI_F32) {
        PAI_LAUNCH(layernorm_bwd_dx_k<float>, dim3((unsigned)M), dim3(256), lds, s, (const float*)dy,
                           (const float*)xs, D, gamma, mean, rstd, (float*)dx);
        if (dgamma_dbeta)
            PAI_LAUNCH(layernorm_bwd_param_k<float>, dim3(cdiv(D, 256), slabs), dim3(256), 0, s,
                               (const float*)dy, (const float*)xs, M, D, mean, rstd, rps, partials);
    } else {
        PAI_LAUNCH(layernorm_bwd_dx_k<bf16_t>, dim3((unsigned)M), dim3(256), lds, s, (const bf16_t*)dy,
                           (const bf16_t*)xs, D, gamma, mean, rstd, (bf16_t*)dx);
        if (dgamma_dbeta)
            PAI_LAUNCH(layernorm_bwd_param_k<bf16_t>, dim3(cdiv(D, 256), slabs), dim3(256), 0, s,
                               (const bf16_t*)dy, (const bf16_t*)xs, M, D, mean, rstd, rps, partials);
    }
    PAI_LAUNCH_CHECK();
    if (dgamma_dbeta) {
        // rows r0 >= M of the last slabs are empty when M < slabs * rps: those blocks still wrote zeros
        return pai_reduce_rows(partials, slabs, 2 * D, dgamma_dbeta, 0, stream);   // [0][D] = dbeta, [1][D] = dgamma
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// GELU (erf form: activation="gelu" of nn.TransformerEncoderLayer, models/trans_unet.py:151-156)
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gelu_k(const T* z, int64_t numel, T* out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < numel; i += (int64_t)gridDim.x * 256) {
        const float v = Conv<T>::ld(z + i);
        Conv<T>::st(out + i, 0.5f * v * (1.0f + erff(v * 0.70710678118654752f)));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gelu_bwd_k(const T* dy, const T* z, int64_t numel, T* dz) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < numel; i += (int64_t)gridDim.x * 256) {
        const float v = Conv<T>::ld(z + i);
        const float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752f));
        const float pdf = 0.3989422804014327f * expf(-0.5f * v * v);
        Conv<T>::st(dz + i, Conv<T>::ld(dy + i) * (cdf + v * pdf));
    }
}

static unsigned ew_blocks(int64_t numel) {
    int64_t b = (numel + 255) / 256;
    return (unsigned)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

extern "C" int pai_gelu(int dtype, const void* z, int64_t numel, void* out, void* stream) {
    PAI_CHECK(dtype == PAI_F32 || dtype == PAI_BF16, "pai_gelu: bad dtype %d", dtype);
    PAI_CHECK(z && out && numel > 0, "pai_gelu: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PAI_F32)
        PAI_LAUNCH(gelu_k<float>, dim3(ew_blocks(numel)), dim3(256), 0, s, (const float*)z, numel, (float*)out);
    else
        PAI_LAUNCH(gelu_k<bf16_t>, dim3(ew_blocks(numel)), dim3(256), 0, s, (const bf16_t*)z, numel, (bf16_t*)out);
    PAI_LAUNCH_CHECK();
    return 0;
}

extern "C" int pai_gelu_bwd(int dtype, const void* dy, const void* z, int64_t numel, void* dz, void* stream) {
    PAI_CHECK(dtype == PAI_F32 || dtype == PAI_BF16, "pai_gelu_bwd: bad dtype %d", dtype);
    PAI_CHECK(dy && z && dz && numel > 0, "pai_gelu_bwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PAI_F32)
        PAI_LAUNCH(gelu_bwd_k<float>, dim3(ew_blocks(numel)), dim3(256), 0, s, (const float*)dy,
                           (const float*)z, numel, (float*)dz);
    else
        PAI_LAUNCH(gelu_bwd_k<bf16_t>, dim3(ew_blocks(numel)), dim3(256), 0, s, (const bf16_t*)dy,
                           (const bf16_t*)z, numel, (bf16_t*)dz);
    PAI_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// Multi-head attention core on packed projections qkv [S*B][3E] (row = s*B + b; q | k | v along the columns,
// head h = columns h*hd .. (h+1)*hd of each): nn.MultiheadAttention inside nn.TransformerEncoderLayer with
// batch_first = False (models/trans_unet.py:151-156,171-175 -- the sequence axis is the image batch).
// One workgroup per (b, h, query i): the S scores live in LDS, the probabilities are kept for the backward pass.
// The whole op is ~70 MFLOP at the benchmark size: it is bound by launch latency, not by any pipe.
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void mha_fwd_k(const T* qkv, int S, int B, int heads, int hd, float scale,
                                                 T* out, float* probs, const float* mask) {
    extern __shared__ float sm[];   // q[hd] | sc[S] | red[8]
    float* q = sm;
    float* sc = sm + hd;
    float* red = sc + S;
    const int i = blockIdx.x % S;
    const int bh = blockIdx.x / S;
    const int h = bh % heads, b = bh / heads;
    const int E = heads * hd;
    const int64_t rs = (int64_t)B * 3 * E;   // stride between sequence positions
    const T* base = qkv + (int64_t)b * 3 * E + h * hd;
    for (int d = threadIdx.x; d < hd; d += 256) q[d] = Conv<T>::ld(base + i * rs + d) * scale;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int j = w; j < S; j += 4) {
        const T* k = base + j * rs + E;
        float a = 0.f;
        for (int d = lane; d < hd; d += 64) a = fmaf(q[d], Conv<T>::ld(k + d), a);
        a = wave_sum(a);
        if (lane == 0) sc[j] = a;
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int j = threadIdx.x; j < S; j += 256) mx = fmaxf(mx, sc[j]);
    mx = wave_max(mx);
    if (lane == 0) red[w] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int j = threadIdx.x; j < S; j += 256) {
        const float e = expf(sc[j] - mx);
        sc[j] = e;
        sum += e;
    }
    sum = block_sum(sum, red + 4);
    const float inv = 1.0f / sum;
    float* pr = probs + ((int64_t)bh * S + i) * S;
    const float* mk = mask ? mask + ((int64_t)bh * S + i) * S : nullptr;   // attention dropout: 0 or 1 / (1 - p)
    for (int j = threadIdx.x; j < S; j += 256) {
        const float p = sc[j] * inv;
        pr[j] = p;                      // the softmax itself is what the backward pass needs
        sc[j] = mk ? p * mk[j] : p;
    }
    __syncthreads();
    T* o = out + ((int64_t)i * B + b) * E + h * hd;
    for (int d = threadIdx.x; d < hd; d += 256) {
        float a = 0.f;
        for (int j = 0; j < S; ++j) a = fmaf(sc[j], Conv<T>::ld(base + j * rs + 2 * E + d), a);
        Conv<T>::st(o + d, a);
    }
}

// per (b, h, i): dP[j] = <dO_i, V_j>, dS[j] = P[j] (dP[j] - sum_j' P[j'] dP[j']), dQ_i = scale sum_j dS[j] K_j
template <typename T>
__global__ __launch_bounds__(256) void mha_bwd_q_k(const T* dout, const T* qkv, const float* probs, int S, int B,
                                                   int heads, int hd, float scale, T* dqkv, float* ds, const float* mask) {
    extern __shared__ float sm[];   // do[hd] | dp[S] | red[8]
    float* dov = sm;
    float* dp = sm + hd;
    float* red = dp + S;
    const int i = blockIdx.x % S;
    const int bh = blockIdx.x / S;
    const int h = bh % heads, b = bh / heads;
    const int E = heads * hd;
    const int64_t rs = (int64_t)B * 3 * E;
    const T* base = qkv + (int64_t)b * 3 * E + h * hd;
    const T* dor = dout + ((int64_t)i * B + b) * E + h * hd;
    for (int d = threadIdx.x; d < hd; d += 256) dov[d] = Conv<T>::ld(dor + d);
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int j = w; j < S; j += 4) {
        const T* v = base + j * rs + 2 * E;
        float a = 0.f;
        for (int d = lane; d < hd; d += 64) a = fmaf(dov[d], Conv<T>::ld(v + d), a);
        a = wave_sum(a);
        if (lane == 0) dp[j] = mask ? a * mask[((int64_t)bh * S + i) * S + j] : a;
    }
    __syncthreads();
    const float* pr = probs + ((int64_t)bh * S + i) * S;
    float dl = 0.f;
    for (int j = threadIdx.x; j < S; j += 256) dl = fmaf(pr[j], dp[j], dl);
    dl = block_sum(dl, red);
    float* dsr = ds + ((int64_t)bh * S + i) * S;
    for (int j = threadIdx.x; j < S; j += 256) {
        const float v = pr[j] * (dp[j] - dl);
        dp[j] = v;
        dsr[j] = v;
    }
    __syncthreads();
    T* dq = dqkv + (int64_t)i * rs + (int64_t)b * 3 * E + h * hd;
    for (int d = threadIdx.x; d < hd; d += 256) {
        float a = 0.f;
        for (int j = 0; j < S; ++j) a = fmaf(dp[j], Conv<T>::ld(base + j * rs + E + d), a);
        Conv<T>::st(dq + d, a * scale);
    }
}

// per (b, h, key j): dV_j = sum_i P[i][j] dO_i,  dK_j = scale sum_i dS[i][j] Q_i
template <typename T>
__global__ __launch_bounds__(256) void mha_bwd_kv_k(const T* dout, const T* qkv, const float* probs, const float* ds,
                                                    int S, int B, int heads, int hd, float scale, T* dqkv, const float* mask) {
    extern __shared__ float sm[];   // p[S] | s[S]
    float* pc = sm;
    float* dc = sm + S;
    const int j = blockIdx.x % S;
    const int bh = blockIdx.x / S;
    const int h = bh % heads, b = bh / heads;
    const int E = heads * hd;
    const int64_t rs = (int64_t)B * 3 * E;
    for (int i = threadIdx.x; i < S; i += 256) {
        const int64_t e = ((int64_t)bh * S + i) * S + j;
        pc[i] = mask ? probs[e] * mask[e] : probs[e];
        dc[i] = ds[e];
    }
    __syncthreads();
    const T* qb = qkv + (int64_t)b * 3 * E + h * hd;
    const T* dob = dout + (int64_t)b * E + h * hd;
    T* dk = dqkv + (int64_t)j * rs + (int64_t)b * 3 * E + E + h * hd;
    T* dv = dk + E;
    for (int d = threadIdx.x; d < hd; d += 256) {
        float av = 0.f, ak = 0.f;
        for (int i = 0; i < S; ++i) {
            av = fmaf(pc[i], Conv<T>::ld(dob + (int64_t)i * B * E + d), av);
            ak = fmaf(dc[i], Conv<T>::ld(qb + i * rs + d), ak);
        }
        Conv<T>::st(dv + d, av);
        Conv<T>::st(dk + d, ak * scale);
    }
}

static int mha_check(const char* who, int dtype, int S, int B, int heads, int hd) {
    PAI_CHECK(dtype == PAI_F32 || dtype == PAI_BF16, "%s: bad dtype %d", who, dtype);
    PAI_CHECK(S > 0 && B > 0 && heads > 0 && hd > 0, "%s: bad shape S=%d B=%d heads=%d hd=%d", who, S, B, heads, hd);
    PAI_CHECK((int64_t)(hd + 2 * S + 16) * 4 <= 64 * 1024, "%s: S=%d, head dim %d do not fit the 64 KB row buffers", who, S, hd);
    PAI_CHECK((int64_t)S * B * heads < ((int64_t)1 << 31), "%s: too many attention rows", who);
    return 0;
}

extern "C" int pai_mha_fwd(int dtype, const void* qkv, int S, int B, int heads, int hd, const float* mask, void* out,
                           float* probs, void* stream) {
    if (mha_check("pai_mha_fwd", dtype, S, B, heads, hd)) return 1;
    PAI_CHECK(qkv && out && probs, "pai_mha_fwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const float scale = 1.0f / sqrtf((float)hd);
    const size_t lds = (size_t)(hd + S + 16) * sizeof(float);
    const dim3 grid((unsigned)(S * B * heads));
    if (dtype == PAI_F32)
        PAI_LAUNCH(mha_fwd_k<float>, grid, dim3(256), lds, s, (const float*)qkv, S, B, heads, hd, scale,
                           (float*)out, probs, mask);
    else
        PAI_LAUNCH(mha_fwd_k<bf16_t>, grid, dim3(256), lds, s, (const bf16_t*)qkv, S, B, heads, hd, scale,
                           (bf16_t*)out, probs, mask);
    PAI_LAUNCH_CHECK();
    return 0;
}

extern "C" int pai_mha_bwd(int dtype, const void* dout, const void* qkv, const float* probs, int S, int B, int heads,
                           int hd, const float* mask, void* dqkv, float* ds_workspace, void* stream) {
    if (mha_check("pai_mha_bwd", dtype, S, B, heads, hd)) return 1;
    PAI_CHECK(dout && qkv && probs && dqkv && ds_workspace, "pai_mha_bwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const float scale = 1.0f / sqrtf((float)hd);
    const size_t lds_q = (size_t)(hd + S + 16) * sizeof(float), lds_kv = (size_t)(2 * S) * sizeof(float);
    const dim3 grid((unsigned)(S * B * heads));
    if (dtype == PAI_F32) {
        PAI_LAUNCH(mha_bwd_q_k<float>, grid, dim3(256), lds_q, s, (const float*)dout, (const float*)qkv, probs,
                           S, B, heads, hd, scale, (float*)dqkv, ds_workspace, mask);
        PAI_LAUNCH(mha_bwd_kv_k<float>, grid, dim3(256), lds_kv, s, (const float*)dout, (const float*)qkv,
                           probs, ds_workspace, S, B, heads, hd, scale, (float*)dqkv, mask);
    } else {
        PAI_LAUNCH(mha_bwd_q_k<bf16_t>, grid, dim3(256), lds_q, s, (const bf16_t*)dout, (const bf16_t*)qkv,
                           probs, S, B, heads, hd, scale, (bf16_t*)dqkv, ds_workspace, mask);
        PAI_LAUNCH(mha_bwd_kv_k<bf16_t>, grid, dim3(256), lds_kv, s, (const bf16_t*)dout, (const bf16_t*)qkv,
                           probs, ds_workspace, S, B, heads, hd, scale, (bf16_t*)dqkv, mask);
    }
    PAI_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// Even-pixel subsample of an NHWC tensor and its adjoint: conv(k3, s2, p1)(x) = subsample(conv(k3, s1, p1)(x)) and
// conv(k1, s2)(x) = conv(k1)(subsample(x))  (the strided convolutions of EncoderBlock, models/trans_unet.py:203-227)
// ---------------------------------------------------------------------------------------------------------
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void subsample2_k(const T* src, int N, int H, int W, int C, T* dst) {
    // forward: src [N][H][W][C] -> dst [N][H/2][W/2][C];  backward: src [N][H/2][W/2][C] -> dst [N][H][W][C]
    const int64_t total = BWD ? (int64_t)N * H * W * C : (int64_t)N * (H / 2) * (W / 2) * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        int64_t p = i / C;
        if (BWD) {
            const int x = (int)(p % W);
            p /= W;
            const int y = (int)(p % H);
            const int n = (int)(p / H);
            float v = 0.f;
            if (!(x & 1) && !(y & 1)) v = Conv<T>::ld(src + (((int64_t)n * (H / 2) + y / 2) * (W / 2) + x / 2) * C + c);
            Conv<T>::st(dst + i, v);
        } else {
            const int x = (int)(p % (W / 2));
            p /= (W / 2);
            const int y = (int)(p % (H / 2));
            const int n = (int)(p / (H / 2));
            dst[i] = src[(((int64_t)n * H + 2 * y) * W + 2 * x) * C + c];
        }
    }
}

static int subsample_launch(const char* who, bool bwd, int dtype, const void* src, int N, int H, int W, int C, void* dst,
                            void* stream) {
    PAI_CHECK(dtype == PAI_F32 || dtype == PAI_BF16, "%s: bad dtype %d", who, dtype);
    PAI_CHECK(src && dst && N > 0 && H > 0 && W > 0 && C > 0 && !(H & 1) && !(W & 1), "%s: bad arguments (H, W even)", who);
    hipStream_t s = (hipStream_t)stream;
    const int64_t total = bwd ? (int64_t)N * H * W * C : (int64_t)N * (H / 2) * (W / 2) * C;
    const dim3 grid(ew_blocks(total));
    if (dtype == PAI_F32) {
        if (bwd) PAI_LAUNCH((subsample2_k<float, true>), grid, dim3(256), 0, s, (const float*)src, N, H, W, C, (float*)dst);
        else PAI_LAUNCH((subsample2_k<float, false>), grid, dim3(256), 0, s, (const float*)src, N, H, W, C, (float*)dst);
    } else {
        if (bwd) PAI_LAUNCH((subsample2_k<bf16_t, true>), grid, dim3(256), 0, s, (const bf16_t*)src, N, H, W, C, (bf16_t*)dst);
        else PAI_LAUNCH((subsample2_k<bf16_t, false>), grid, dim3(256), 0, s, (const bf16_t*)src, N, H, W, C, (bf16_t*)dst);
    }
    PAI_LAUNCH_CHECK();
    return 0;
}

extern "C" int pai_subsample2(int dtype, const void* x, int N, int H, int W, int C, void* out, void* stream) {
    return subsample_launch("pai_subsample2", false, dtype, x, N, H, W, C, out, stream);
}
extern "C" int pai_subsample2_bwd(int dtype, const void* dout, int N, int H, int W, int C, void* dx, void* stream) {
    return subsample_launch("pai_subsample2_bwd", true, dtype, dout, N, H, W, C, dx, stream);
}

// ---------------------------------------------------------------------------------------------------------
// BatchNorm partial statistics of a stored tensor [M][C]: rows [slab][2][C] = (sum, sum of squares) of 64-row slabs,
// in the layout pai_bn_finalize reduces (it sums the rows in fp64).
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_k(const T* z, int64_t M, int C, int64_t rows_per_slab, float* stats) {
    __shared__ float rs[256], rq[256];
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_slab, r1 = min(M, r0 + rows_per_slab);
    float* dst = stats + (int64_t)blockIdx.x * 2 * C;
    for (int c0 = 0; c0 < C; c0 += 256) {
        const int width = min(C - c0, 256);
        int lanes = 1;
        while (lanes * 2 * width <= 256) lanes *= 2;
        const int c = tid % width, rl = tid / width;
        float s = 0.f, q = 0.f;
        if (rl < lanes)
            for (int64_t r = r0 + rl; r < r1; r += lanes) {
                const float v = Conv<T>::ld(z + r * C + c0 + c);
                s += v;
                q = fmaf(v, v, q);
            }
        rs[tid] = (rl < lanes) ? s : 0.f;
        rq[tid] = (rl < lanes) ? q : 0.f;
        __syncthreads();
        if (tid < width) {
            float ts = 0.f, tq = 0.f;
            for (int l = 0; l < lanes; ++l) {
                ts += rs[tid + l * width];
                tq += rq[tid + l * width];
            }
            dst[c0 + tid] = ts;
            dst[C + c0 + tid] = tq;
        }
        __syncthreads();
    }
}

static int64_t bn_stats_rps(int64_t M) {
    int64_t rps = 64;
    while ((M + rps - 1) / rps > 4096) rps *= 2;
    return rps;
}

extern "C" int pai_bn_stats_rows(int64_t M) {
    const int64_t rps = bn_stats_rps(M);
    return (int)((M + rps - 1) / rps);
}

extern "C" int pai_bn_stats(int dtype, const void* z, int64_t M, int C, float* stats, void* stream) {
    PAI_CHECK(dtype == PAI_F32 || dtype == PAI_BF16, "pai_bn_stats: bad dtype %d", dtype);
    PAI_CHECK(z && stats && M > 0 && C > 0, "pai_bn_stats: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int64_t rps = bn_stats_rps(M);
    const dim3 grid((unsigned)((M + rps - 1) / rps));
    if (dtype == PAI_F32)
        PAI_LAUNCH(bn_stats_k<float>, grid, dim3(256), 0, s, (const float*)z, M, C, rps, stats);
    else
        PAI_LAUNCH(bn_stats_k<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)z, M, C, rps, stats);
    PAI_LAUNCH_CHECK();
    return 0;
}

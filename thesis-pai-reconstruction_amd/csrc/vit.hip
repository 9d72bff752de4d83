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

// ---------------------------------------------------------------------------------------------------------
// The same attention core on the matrix cores (bf16 storage, S <= 32 tokens per sequence -- the sequence axis is the image
// batch, 32 at BASELINE configs[4] -- head dim a multiple of 32): one workgroup of four waves per (b, h).
//   forward   S^T = K Q^T          v_mfma_f32_32x32x16_bf16, A = K rows, B = Q rows (both contiguous 16-B fragments straight
//                                  from memory), the head dim split over the waves, partial tiles summed through LDS;
//             softmax              one thread per (query, 4 keys), row max / sum over 8 lanes; fp32 probabilities kept for
//                                  the backward pass as before;
//             O = P V              A = P (bf16, LDS rows), B = V fetched from a row-major LDS image with
//                                  ds_read_b64_tr_b16 (the hardware transpose: V is k-strided for this product).
//   backward  dP = dO V^T          A = dO rows (LDS), B = V rows (memory): no transpose;
//             dS = P (dP - <P, dP>) as above;
//             dQ = dS K, dK = dS^T Q, dV = P^T dO    A from small bf16 LDS tiles (dS, dS^T, P^T), B = K, Q, dO through
//                                  transposed reads of their row-major LDS images.
// Rows beyond S are zero fragments (never loaded), keys beyond S get a score of -inf.  Everything else (fp32 storage,
// longer sequences, odd head dims) stays on the vector-ALU kernels above.
// ---------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 mbf8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 mbf4_t;
typedef __attribute__((ext_vector_type(16))) float mf16_t;

__device__ __forceinline__ mbf8_t mha_tr_frag(const bf16_t* img, int ld, int R0, int C0, int lane) {
    // B operand fragment: element e of lane (r = lane & 31, hh = lane >> 5) = img[R0 + e][C0' + (lane & 15)], C0' = the
    // caller's C0 + 16 for the odd 16-lane groups; two transposed reads of 4 rows x 16 columns each (guide T10)
    const int q = (lane & 15) >> 2, p = lane & 3;
    const bf16_t* a0 = img + (size_t)(R0 + q) * ld + C0 + 16 * ((lane >> 4) & 1) + 4 * p;
    typedef mbf4_t __attribute__((address_space(3))) * lds4_t;
    const mbf4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(const void*)a0);
    const mbf4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(const void*)(a0 + 4 * ld));
    return (mbf8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

__device__ __forceinline__ void mha_stage_rows(bf16_t* dst, const bf16_t* src, int64_t rs, int S, int hd, int tid) {
    // 32 rows x hd bf16, row-major; rows >= S are zero
    const int cpr = hd / 8;
    for (int c = tid; c < 32 * cpr; c += 256) {
        const int row = c / cpr, col = (c - row * cpr) * 8;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row < S) v = *(const uint4*)(src + row * rs + col);
        *(uint4*)(dst + (size_t)row * hd + col) = v;
    }
}

__global__ __launch_bounds__(256) void mha_fwd_mfma_k(const bf16_t* qkv, int S, int B, int heads, int hd, float scale,
                                                      bf16_t* out, float* probs, const float* mask) {
    extern __shared__ __attribute__((aligned(16))) unsigned char msm[];
    bf16_t* Vs = (bf16_t*)msm;                                   // [32][hd]
    float* part = (float*)(msm + (size_t)32 * hd * 2);           // [4][32 j][33]
    bf16_t* Ps = (bf16_t*)(part + 4 * 32 * 33);                  // [32 i][32 j]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int bh = blockIdx.x, h = bh % heads, b = bh / heads;
    const int E = heads * hd;
    const int64_t rs = (int64_t)B * 3 * E;
    const bf16_t* base = qkv + (int64_t)b * 3 * E + h * hd;
    mha_stage_rows(Vs, base + 2 * E, rs, S, hd, tid);
    mf16_t acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const mbf8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int ks = w; ks < hd / 16; ks += 4) {
        mbf8_t kf = zero8, qf = zero8;
        if (r < S) {
            kf = *(const mbf8_t*)(base + r * rs + E + 16 * ks + 8 * hh);
            qf = *(const mbf8_t*)(base + r * rs + 16 * ks + 8 * hh);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf, acc, 0, 0, 0);     // D[row j][col i]
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) part[(w * 32 + 8 * (i >> 2) + 4 * hh + (i & 3)) * 33 + r] = acc[i];
    __syncthreads();
    {
        const int i = tid >> 3, j0 = (tid & 7) * 4;
        float sc[4], mx = -INFINITY;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = j0 + e;
            float v = part[(0 * 32 + j) * 33 + i] + part[(1 * 32 + j) * 33 + i] + part[(2 * 32 + j) * 33 + i] + part[(3 * 32 + j) * 33 + i];
            sc[e] = j < S ? v * scale : -INFINITY;
            mx = fmaxf(mx, sc[e]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 4, 64));
        float sum = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) { sc[e] = (j0 + e) < S ? expf(sc[e] - mx) : 0.f; sum += sc[e]; }
        sum += __shfl_xor(sum, 1, 64);
        sum += __shfl_xor(sum, 2, 64);
        sum += __shfl_xor(sum, 4, 64);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = j0 + e;
            float pv = sc[e] * inv;
            if (i < S && j < S) {
                const int64_t idx = ((int64_t)bh * S + i) * S + j;
                probs[idx] = pv;                       // the softmax itself is what the backward pass needs
                if (mask) pv *= mask[idx];             // attention dropout: 0 or 1 / (1 - p)
            } else {
                pv = 0.f;
            }
            Ps[i * 32 + j] = f2bf(pv);
        }
    }
    __syncthreads();
    for (int t = w; t < hd / 32; t += 4) {
        const int d0 = 32 * t;
        mf16_t o;
#pragma unroll
        for (int i = 0; i < 16; ++i) o[i] = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const mbf8_t pf = *(const mbf8_t*)(Ps + r * 32 + 16 * s2 + 8 * hh);
            const mbf8_t vf = mha_tr_frag(Vs, hd, 16 * s2 + 8 * hh, d0, lane);
            o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf, vf, o, 0, 0, 0);      // D[row i][col d]
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int qi = 8 * (i >> 2) + 4 * hh + (i & 3);
            if (qi < S) out[((int64_t)qi * B + b) * E + h * hd + d0 + r] = f2bf(o[i]);
        }
    }
}

__global__ __launch_bounds__(256) void mha_bwd_mfma_k(const bf16_t* dout, const bf16_t* qkv, const float* probs, int S, int B,
                                                      int heads, int hd, float scale, bf16_t* dqkv, const float* mask) {
    extern __shared__ __attribute__((aligned(16))) unsigned char msm[];
    bf16_t* Qs = (bf16_t*)msm;                                   // [32][hd] each, row-major
    bf16_t* Ks = Qs + (size_t)32 * hd;
    bf16_t* Os = Ks + (size_t)32 * hd;                           // dO
    float* part = (float*)(Os + (size_t)32 * hd);                // [4][32 i][33]
    bf16_t* dSs = (bf16_t*)(part + 4 * 32 * 33);                 // [32 i][32 j]
    bf16_t* dSt = dSs + 32 * 32;                                 // [32 j][32 i]
    bf16_t* Pt = dSt + 32 * 32;                                  // [32 j][32 i]  (softmax x dropout mask)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int bh = blockIdx.x, h = bh % heads, b = bh / heads;
    const int E = heads * hd;
    const int64_t rs = (int64_t)B * 3 * E, ors = (int64_t)B * E;
    const bf16_t* base = qkv + (int64_t)b * 3 * E + h * hd;
    mha_stage_rows(Qs, base, rs, S, hd, tid);
    mha_stage_rows(Ks, base + E, rs, S, hd, tid);
    mha_stage_rows(Os, dout + (int64_t)b * E + h * hd, ors, S, hd, tid);
    __syncthreads();
    mf16_t acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const mbf8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int ks = w; ks < hd / 16; ks += 4) {
        const mbf8_t of = *(const mbf8_t*)(Os + (size_t)r * hd + 16 * ks + 8 * hh);
        mbf8_t vf = zero8;
        if (r < S) vf = *(const mbf8_t*)(base + r * rs + 2 * E + 16 * ks + 8 * hh);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(of, vf, acc, 0, 0, 0);     // dP: D[row i][col j]
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) part[(w * 32 + 8 * (i >> 2) + 4 * hh + (i & 3)) * 33 + r] = acc[i];
    __syncthreads();
    {
        const int i = tid >> 3, j0 = (tid & 7) * 4;
        float dp[4], pr[4], pm[4], dl = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = j0 + e;
            dp[e] = part[(0 * 32 + i) * 33 + j] + part[(1 * 32 + i) * 33 + j] + part[(2 * 32 + i) * 33 + j] + part[(3 * 32 + i) * 33 + j];
            pr[e] = 0.f;
            pm[e] = 0.f;
            if (i < S && j < S) {
                const int64_t idx = ((int64_t)bh * S + i) * S + j;
                pr[e] = probs[idx];
                const float mk = mask ? mask[idx] : 1.f;
                pm[e] = pr[e] * mk;
                dp[e] *= mk;
            } else {
                dp[e] = 0.f;
            }
            dl = fmaf(pr[e], dp[e], dl);
        }
        dl += __shfl_xor(dl, 1, 64);
        dl += __shfl_xor(dl, 2, 64);
        dl += __shfl_xor(dl, 4, 64);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = j0 + e;
            const bf16_t dsv = f2bf(pr[e] * (dp[e] - dl));
            dSs[i * 32 + j] = dsv;
            dSt[j * 32 + i] = dsv;
            Pt[j * 32 + i] = f2bf(pm[e]);
        }
    }
    __syncthreads();
    bf16_t* drow = dqkv + (int64_t)b * 3 * E + h * hd;
    for (int t = w; t < hd / 32; t += 4) {
        const int d0 = 32 * t;
        mf16_t aq, ak, av;
#pragma unroll
        for (int i = 0; i < 16; ++i) aq[i] = ak[i] = av[i] = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int k0 = 16 * s2 + 8 * hh;
            const mbf8_t dsf = *(const mbf8_t*)(dSs + r * 32 + k0);      // A[row i][k = j]
            const mbf8_t dtf = *(const mbf8_t*)(dSt + r * 32 + k0);      // A[row j][k = i]
            const mbf8_t ptf = *(const mbf8_t*)(Pt + r * 32 + k0);       // A[row j][k = i]
            aq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dsf, mha_tr_frag(Ks, hd, k0, d0, lane), aq, 0, 0, 0);
            ak = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dtf, mha_tr_frag(Qs, hd, k0, d0, lane), ak, 0, 0, 0);
            av = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ptf, mha_tr_frag(Os, hd, k0, d0, lane), av, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = 8 * (i >> 2) + 4 * hh + (i & 3);
            if (row < S) {
                bf16_t* o = drow + (int64_t)row * rs + d0 + r;
                o[0] = f2bf(aq[i] * scale);
                o[E] = f2bf(ak[i] * scale);
                o[2 * E] = f2bf(av[i]);
            }
        }
    }
}

static bool mha_mfma_ok(int dtype, int S, int hd) {
    return dtype == PAI_BF16 && S <= 32 && (hd % 32) == 0 && hd <= 512 && pai_tunable("mha_mfma", 1);
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
    if (mha_mfma_ok(dtype, S, hd)) {
        const size_t ml = (size_t)32 * hd * 2 + 4 * 32 * 33 * sizeof(float) + 32 * 32 * 2;
        static PerDeviceOnce attr;
        if (attr.first()) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mha_fwd_mfma_k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e == hipSuccess)
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mha_bwd_mfma_k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            PAI_CHECK(e == hipSuccess, "hipFuncSetAttribute(max dynamic LDS): %s", hipGetErrorString(e));
        }
        PAI_LAUNCH(mha_fwd_mfma_k, dim3((unsigned)(B * heads)), dim3(256), ml, s, (const bf16_t*)qkv, S, B, heads, hd, scale,
                   (bf16_t*)out, probs, mask);
        PAI_LAUNCH_CHECK();
        return 0;
    }
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
    if (mha_mfma_ok(dtype, S, hd)) {
        const size_t ml = (size_t)3 * 32 * hd * 2 + 4 * 32 * 33 * sizeof(float) + 3 * 32 * 32 * 2;
        static PerDeviceOnce attr;
        if (attr.first()) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mha_bwd_mfma_k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            PAI_CHECK(e == hipSuccess, "hipFuncSetAttribute(max dynamic LDS): %s", hipGetErrorString(e));
        }
        PAI_LAUNCH(mha_bwd_mfma_k, dim3((unsigned)(B * heads)), dim3(256), ml, s, (const bf16_t*)dout, (const bf16_t*)qkv, probs,
                   S, B, heads, hd, scale, (bf16_t*)dqkv, mask);
        PAI_LAUNCH_CHECK();
        return 0;
    }
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

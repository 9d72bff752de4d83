// Streaming forms of the big bf16 tensor passes: BatchNorm apply / backward reduce / backward apply, the residual-block
// tail and the residual sum (reference models/res_unet.py:133-171 -- every nn.BatchNorm2d / ReLU / `+` of a block -- and
// models/pix2pix.py:70,106) for tensors of more than 4096 rows.
//
// Why a second form: the generic kernels of bn.hip / resnet.hip ran the residual U-Net's 1 GB passes (configs[3], 512 x 512,
// 128 channels) at 3.6-3.7 TB/s, scripts/micro/hbm_pass.hip reaches 6.0-6.8 TB/s with the same bytes.  The difference is
// not the memory system: per 16-byte vector those kernels issue a 64-bit modulo, seven 32-byte parameter loads, run-time
// activation switches per element (v_cmp / v_cndmask chains) and keep ONE vector per tensor in flight.  Here
//   * the channel group of a thread is loop-invariant (C / 8 is a power of two <= 256 and every stride is a multiple of it),
//     so the per-channel coefficients are loaded and combined ONCE per thread;
//   * the BatchNorm backward is dz = A du + (B (z - mean) + K) with A = gamma rstd, B = -A rstd sum(du xhat) / M,
//     K = -A sum(du) / M: two fused multiply-adds per element;
//   * the activation is a template parameter;
//   * four vectors per tensor are in flight per thread, tensors of >= 128 MB move with non-temporal loads and stores.
// fp32 (the parity mode), small layers (<= 4096 rows: bit-identical to the one-launch forms of bn.hip) and odd channel counts
// stay on the generic kernels; `pai_set_tunable("ew_stream", 0)` sends everything there (A/B, tests).
#include "common.h"

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int SV = 4;             // 16-byte vectors in flight per thread and tensor

template <bool NT> __device__ __forceinline__ u32x4 ldv(const u32x4* p) {
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
}
template <bool NT> __device__ __forceinline__ void stv(u32x4* p, u32x4 v) {
    if (NT) __builtin_nontemporal_store(v, p); else *p = v;
}
__device__ __forceinline__ void unpack8(u32x4 w, float* o) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        o[2 * i] = __uint_as_float(w[i] << 16);
        o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
}
__device__ __forceinline__ u32x4 pack8(const float* v) {
    u32x4 w;
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = pk2bf(v[2 * i], v[2 * i + 1]);
    return w;
}
template <int ACT> __device__ __forceinline__ float actc(float v) {
    if (ACT == PAI_ACT_RELU) return v > 0.f ? v : 0.f;
    if (ACT == PAI_ACT_LRELU) return v > 0.f ? v : 0.2f * v;
    return v;
}
// g * act'(pre), as the generic kernels form it (rounded to the storage type where the product is not exact)
template <int ACT> __device__ __forceinline__ float actg(float pre, float g) {
    if (ACT == PAI_ACT_RELU) return pre > 0.f ? g : 0.f;
    if (ACT == PAI_ACT_LRELU) return pre > 0.f ? g : bf2f(f2bf(0.2f * g));
    return g;
}
__device__ __forceinline__ void ld8(const float* p, float* o) {
    const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}

// ---- out = act(z * scale + shift): bit-identical to bn_apply_k ------------------------------------------------------------
template <int ACT, bool NT>
__global__ __launch_bounds__(256) void bn_apply_stream_k(const u32x4* z, int64_t nvec, int G, const float* scale,
                                                         const float* shift, u32x4* out) {
    const int cg = threadIdx.x & (G - 1);
    float sc[8], sh[8];
    ld8(scale + cg * 8, sc);
    ld8(shift + cg * 8, sh);
    const int64_t step = (int64_t)gridDim.x * 256 * SV;
    for (int64_t i = (int64_t)blockIdx.x * 256 * SV + threadIdx.x; i < nvec; i += step) {
        u32x4 w[SV];
#pragma unroll
        for (int k = 0; k < SV; ++k)
            if (i + k * 256 < nvec) w[k] = ldv<NT>(z + i + k * 256);
#pragma unroll
        for (int k = 0; k < SV; ++k)
            if (i + k * 256 < nvec) {
                float v[8];
                unpack8(w[k], v);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = actc<ACT>(fmaf(v[e], sc[e], sh[e]));
                stv<NT>(out + i + k * 256, pack8(v));
            }
    }
}

// ---- out = act(act_a(za * sca + sha) + (zb * scb + shb | zb)): bit-identical to bn2_add_act_k -----------------------------
template <int ACT_A, int ACT, bool AFFB, bool NT>
__global__ __launch_bounds__(256) void bn2_add_act_stream_k(const u32x4* za, const float* sca, const float* sha,
                                                            const u32x4* zb, const float* scb, const float* shb,
                                                            int64_t nvec, int G, u32x4* out) {
    const int cg = threadIdx.x & (G - 1);
    float s1[8], h1[8], s2[8], h2[8];
    ld8(sca + cg * 8, s1);
    ld8(sha + cg * 8, h1);
    if (AFFB) { ld8(scb + cg * 8, s2); ld8(shb + cg * 8, h2); }
    const int64_t step = (int64_t)gridDim.x * 256 * SV;
    for (int64_t i = (int64_t)blockIdx.x * 256 * SV + threadIdx.x; i < nvec; i += step) {
        u32x4 wa[SV], wb[SV];
#pragma unroll
        for (int k = 0; k < SV; ++k)
            if (i + k * 256 < nvec) { wa[k] = ldv<NT>(za + i + k * 256); wb[k] = ldv<NT>(zb + i + k * 256); }
#pragma unroll
        for (int k = 0; k < SV; ++k)
            if (i + k * 256 < nvec) {
                float a[8], b[8];
                unpack8(wa[k], a);
                unpack8(wb[k], b);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float x = actc<ACT_A>(fmaf(a[e], s1[e], h1[e]));
                    const float y = AFFB ? fmaf(b[e], s2[e], h2[e]) : b[e];
                    a[e] = actc<ACT>(x + y);
                }
                stv<NT>(out + i + k * 256, pack8(a));
            }
    }
}

// ---- out = act(a + b): bit-identical to add_act_k ----------------------------------------------------------------------------
template <int ACT, bool NT>
__global__ __launch_bounds__(256) void add_act_stream_k(const u32x4* a, const u32x4* b, int64_t nvec, u32x4* out) {
    const int64_t step = (int64_t)gridDim.x * 256 * SV;
    for (int64_t i = (int64_t)blockIdx.x * 256 * SV + threadIdx.x; i < nvec; i += step) {
        u32x4 wa[SV], wb[SV];
#pragma unroll
        for (int k = 0; k < SV; ++k)
            if (i + k * 256 < nvec) { wa[k] = ldv<NT>(a + i + k * 256); wb[k] = ldv<NT>(b + i + k * 256); }
#pragma unroll
        for (int k = 0; k < SV; ++k)
            if (i + k * 256 < nvec) {
                float x[8], y[8];
                unpack8(wa[k], x);
                unpack8(wb[k], y);
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = actc<ACT>(x[e] + y[e]);
                stv<NT>(out + i + k * 256, pack8(x));
            }
    }
}

// ---- BatchNorm backward, pass 1: partial sums of du = g * act'(z * scale + shift) and du * xhat per slab of rows ------------
// Same slab-per-block partial rows as bn_bwd_reduce_k ([block][0] = sum du, [block][1] = sum du * xhat); du is not stored.
// sum du * xhat is formed as rstd * sum du * (z - mean).
template <int ACT, bool NT>
__global__ __launch_bounds__(256) void bn_bwd_reduce_stream_k(const u32x4* g, const u32x4* z, int64_t M, int G,
                                                              int64_t rows_per_block, const float* mean, const float* rstd,
                                                              const float* scale, const float* shift, float* partials) {
    __shared__ float red[2][256][8];
    const int tid = threadIdx.x, cg = tid & (G - 1);
    float mu[8], sc[8], sh[8], s1[8], s2[8];
    ld8(mean + cg * 8, mu);
    if (ACT != PAI_ACT_NONE) { ld8(scale + cg * 8, sc); ld8(shift + cg * 8, sh); }
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(M, r0 + rows_per_block);
    const int64_t i1 = r1 * G;
    for (int64_t i = r0 * G + tid; i < i1; i += 256 * SV) {
        u32x4 wg[SV], wz[SV];
#pragma unroll
        for (int k = 0; k < SV; ++k)
            if (i + k * 256 < i1) { wg[k] = ldv<NT>(g + i + k * 256); wz[k] = ldv<NT>(z + i + k * 256); }
#pragma unroll
        for (int k = 0; k < SV; ++k)
            if (i + k * 256 < i1) {
                float gv[8], zv[8];
                unpack8(wg[k], gv);
                unpack8(wz[k], zv);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = ACT == PAI_ACT_NONE ? gv[e] : actg<ACT>(fmaf(zv[e], sc[e], sh[e]), gv[e]);
                    s1[e] += d;
                    s2[e] = fmaf(d, zv[e] - mu[e], s2[e]);
                }
            }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[0][tid][e] = s1[e]; red[1][tid][e] = s2[e]; }
    __syncthreads();
    if (tid < G) {
        const int lanes = 256 / G, C = G * 8;
        float rs[8];
        ld8(rstd + cg * 8, rs);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float t1 = 0.f, t2 = 0.f;
            for (int l = 0; l < lanes; ++l) { t1 += red[0][tid + l * G][e]; t2 += red[1][tid + l * G][e]; }
            partials[((size_t)blockIdx.x * 2 + 0) * C + cg * 8 + e] = t1;
            partials[((size_t)blockIdx.x * 2 + 1) * C + cg * 8 + e] = t2 * rs[e];
        }
    }
}

// ---- BatchNorm backward, pass 2: dz = A du + (B (z - mean) + K), du rebuilt from g as pass 1 formed it ----------------------
template <int ACT, bool NT>
__global__ __launch_bounds__(256) void bn_bwd_apply_stream_k(const u32x4* g, const u32x4* z, int64_t nvec, int G, float inv_m,
                                                             const float* mean, const float* rstd, const float* gamma,
                                                             const float* sums, const float* scale, const float* shift,
                                                             u32x4* dz) {
    const int cg = threadIdx.x & (G - 1), C = G * 8;
    float A[8], B[8], K[8], mu[8], sc[8], sh[8];
    {
        float rs[8], gm[8], sb[8], sg[8];
        ld8(mean + cg * 8, mu);
        ld8(rstd + cg * 8, rs);
        ld8(sums + cg * 8, sb);
        ld8(sums + C + cg * 8, sg);
        if (gamma) ld8(gamma + cg * 8, gm);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            A[e] = (gamma ? gm[e] : 1.f) * rs[e];
            B[e] = -A[e] * rs[e] * (sg[e] * inv_m);
            K[e] = -A[e] * (sb[e] * inv_m);
        }
    }
    if (ACT != PAI_ACT_NONE) { ld8(scale + cg * 8, sc); ld8(shift + cg * 8, sh); }
    const int64_t step = (int64_t)gridDim.x * 256 * SV;
    for (int64_t i = (int64_t)blockIdx.x * 256 * SV + threadIdx.x; i < nvec; i += step) {
        u32x4 wg[SV], wz[SV];
#pragma unroll
        for (int k = 0; k < SV; ++k)
            if (i + k * 256 < nvec) { wg[k] = ldv<NT>(g + i + k * 256); wz[k] = ldv<NT>(z + i + k * 256); }
#pragma unroll
        for (int k = 0; k < SV; ++k)
            if (i + k * 256 < nvec) {
                float gv[8], zv[8];
                unpack8(wg[k], gv);
                unpack8(wz[k], zv);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = ACT == PAI_ACT_NONE ? gv[e] : actg<ACT>(fmaf(zv[e], sc[e], sh[e]), gv[e]);
                    gv[e] = fmaf(A[e], d, fmaf(B[e], zv[e] - mu[e], K[e]));
                }
                stv<NT>(dz + i + k * 256, pack8(gv));
            }
    }
}

// ---- the tail of a residual block, backward: BOTH BatchNorms (residual branch with its activation, skip branch without) read
// the same incoming gradient d -- one pass over (d, za, zb) for the two sets of partial sums, one for the two dz
template <int ACT, bool NT>
__global__ __launch_bounds__(256) void bn2_bwd_reduce_stream_k(const u32x4* d, const u32x4* za, const u32x4* zb, int64_t M, int G,
                                                               int64_t rows_per_block, const float* mean_a, const float* rstd_a,
                                                               const float* scale_a, const float* shift_a, const float* mean_b,
                                                               const float* rstd_b, float* part_a, float* part_b) {
    __shared__ float red[4][256][8];
    const int tid = threadIdx.x, cg = tid & (G - 1);
    float mua[8], mub[8], sc[8], sh[8], a1[8], a2[8], b1[8], b2[8];
    ld8(mean_a + cg * 8, mua);
    ld8(mean_b + cg * 8, mub);
    if (ACT != PAI_ACT_NONE) { ld8(scale_a + cg * 8, sc); ld8(shift_a + cg * 8, sh); }
#pragma unroll
    for (int e = 0; e < 8; ++e) a1[e] = a2[e] = b1[e] = b2[e] = 0.f;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(M, r0 + rows_per_block);
    const int64_t i1 = r1 * G;
    constexpr int V = 2;           // three tensors: two vectors each in flight
    for (int64_t i = r0 * G + tid; i < i1; i += 256 * V) {
        u32x4 wd[V], wa[V], wb[V];
#pragma unroll
        for (int k = 0; k < V; ++k)
            if (i + k * 256 < i1) { wd[k] = ldv<NT>(d + i + k * 256); wa[k] = ldv<NT>(za + i + k * 256); wb[k] = ldv<NT>(zb + i + k * 256); }
#pragma unroll
        for (int k = 0; k < V; ++k)
            if (i + k * 256 < i1) {
                float dv[8], av[8], bv[8];
                unpack8(wd[k], dv);
                unpack8(wa[k], av);
                unpack8(wb[k], bv);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float du = ACT == PAI_ACT_NONE ? dv[e] : actg<ACT>(fmaf(av[e], sc[e], sh[e]), dv[e]);
                    a1[e] += du;
                    a2[e] = fmaf(du, av[e] - mua[e], a2[e]);
                    b1[e] += dv[e];
                    b2[e] = fmaf(dv[e], bv[e] - mub[e], b2[e]);
                }
            }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[0][tid][e] = a1[e]; red[1][tid][e] = a2[e]; red[2][tid][e] = b1[e]; red[3][tid][e] = b2[e]; }
    __syncthreads();
    if (tid < G) {
        const int lanes = 256 / G, C = G * 8;
        float ra[8], rb[8];
        ld8(rstd_a + cg * 8, ra);
        ld8(rstd_b + cg * 8, rb);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float t[4] = {0.f, 0.f, 0.f, 0.f};
            for (int l = 0; l < lanes; ++l)
#pragma unroll
                for (int q = 0; q < 4; ++q) t[q] += red[q][tid + l * G][e];
            part_a[((size_t)blockIdx.x * 2 + 0) * C + cg * 8 + e] = t[0];
            part_a[((size_t)blockIdx.x * 2 + 1) * C + cg * 8 + e] = t[1] * ra[e];
            part_b[((size_t)blockIdx.x * 2 + 0) * C + cg * 8 + e] = t[2];
            part_b[((size_t)blockIdx.x * 2 + 1) * C + cg * 8 + e] = t[3] * rb[e];
        }
    }
}

template <int ACT, bool NT>
__global__ __launch_bounds__(256) void bn2_bwd_apply_stream_k(const u32x4* d, const u32x4* za, const u32x4* zb, int64_t nvec, int G,
                                                              float inv_m, const float* mean_a, const float* rstd_a,
                                                              const float* gamma_a, const float* sums_a, const float* scale_a,
                                                              const float* shift_a, const float* mean_b, const float* rstd_b,
                                                              const float* gamma_b, const float* sums_b, u32x4* dza, u32x4* dzb) {
    const int cg = threadIdx.x & (G - 1), C = G * 8;
    float Aa[8], Ba[8], Ka[8], mua[8], Ab[8], Bb[8], Kb[8], mub[8], sc[8], sh[8];
    {
        float rs[8], gm[8], sb[8], sg[8];
        ld8(mean_a + cg * 8, mua); ld8(rstd_a + cg * 8, rs); ld8(sums_a + cg * 8, sb); ld8(sums_a + C + cg * 8, sg);
        if (gamma_a) ld8(gamma_a + cg * 8, gm);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            Aa[e] = (gamma_a ? gm[e] : 1.f) * rs[e];
            Ba[e] = -Aa[e] * rs[e] * (sg[e] * inv_m);
            Ka[e] = -Aa[e] * (sb[e] * inv_m);
        }
        ld8(mean_b + cg * 8, mub); ld8(rstd_b + cg * 8, rs); ld8(sums_b + cg * 8, sb); ld8(sums_b + C + cg * 8, sg);
        if (gamma_b) ld8(gamma_b + cg * 8, gm);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            Ab[e] = (gamma_b ? gm[e] : 1.f) * rs[e];
            Bb[e] = -Ab[e] * rs[e] * (sg[e] * inv_m);
            Kb[e] = -Ab[e] * (sb[e] * inv_m);
        }
    }
    if (ACT != PAI_ACT_NONE) { ld8(scale_a + cg * 8, sc); ld8(shift_a + cg * 8, sh); }
    constexpr int V = 2;
    const int64_t step = (int64_t)gridDim.x * 256 * V;
    for (int64_t i = (int64_t)blockIdx.x * 256 * V + threadIdx.x; i < nvec; i += step) {
        u32x4 wd[V], wa[V], wb[V];
#pragma unroll
        for (int k = 0; k < V; ++k)
            if (i + k * 256 < nvec) { wd[k] = ldv<NT>(d + i + k * 256); wa[k] = ldv<NT>(za + i + k * 256); wb[k] = ldv<NT>(zb + i + k * 256); }
#pragma unroll
        for (int k = 0; k < V; ++k)
            if (i + k * 256 < nvec) {
                float dv[8], av[8], bv[8], oa[8], ob[8];
                unpack8(wd[k], dv);
                unpack8(wa[k], av);
                unpack8(wb[k], bv);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float du = ACT == PAI_ACT_NONE ? dv[e] : actg<ACT>(fmaf(av[e], sc[e], sh[e]), dv[e]);
                    oa[e] = fmaf(Aa[e], du, fmaf(Ba[e], av[e] - mua[e], Ka[e]));
                    ob[e] = fmaf(Ab[e], dv[e], fmaf(Bb[e], bv[e] - mub[e], Kb[e]));
                }
                stv<NT>(dza + i + k * 256, pack8(oa));
                stv<NT>(dzb + i + k * 256, pack8(ob));
            }
    }
}

// ---- host side --------------------------------------------------------------------------------------------------------------
// which calls take the streaming form: bf16, more than 4096 rows (the small layers keep the arithmetic of their one-launch
// forms), at least 32768 vectors, C / 8 a power of two <= 256
static bool stream_ok(int dtype, int64_t M, int C) {
    if (dtype != PAI_BF16 || C % 8 != 0 || M <= 4096) return false;
    const int G = C / 8;
    if (G < 1 || G > 256 || (G & (G - 1)) != 0 || M * G < (1 << 15)) return false;
    return pai_tunable("ew_stream", 1) != 0;
}
static bool stream_nt(int64_t nvec) { return nvec * 16 >= (int64_t)pai_tunable("ew_stream_nt_mb", 128) << 20; }
static int sweep_grid(int64_t nvec) {
    int64_t b = (nvec + 256 * SV - 1) / (256 * SV);
    const int64_t cap = pai_tunable("ew_stream_blocks", 16384);
    if (b > cap) b = cap;
    return b < 1 ? 1 : (int)b;
}
static bool act3(int act) { return act == PAI_ACT_NONE || act == PAI_ACT_RELU || act == PAI_ACT_LRELU; }

#define EW_ACT_NT(KERN, act, nt, grid, s, ...)                                                                    \
    do {                                                                                                          \
        if (nt) {                                                                                                 \
            if (act == PAI_ACT_RELU) PAI_LAUNCH((KERN<PAI_ACT_RELU, true>), grid, dim3(256), 0, s, __VA_ARGS__);   \
            else if (act == PAI_ACT_LRELU) PAI_LAUNCH((KERN<PAI_ACT_LRELU, true>), grid, dim3(256), 0, s, __VA_ARGS__); \
            else PAI_LAUNCH((KERN<PAI_ACT_NONE, true>), grid, dim3(256), 0, s, __VA_ARGS__);                       \
        } else {                                                                                                  \
            if (act == PAI_ACT_RELU) PAI_LAUNCH((KERN<PAI_ACT_RELU, false>), grid, dim3(256), 0, s, __VA_ARGS__);  \
            else if (act == PAI_ACT_LRELU) PAI_LAUNCH((KERN<PAI_ACT_LRELU, false>), grid, dim3(256), 0, s, __VA_ARGS__); \
            else PAI_LAUNCH((KERN<PAI_ACT_NONE, false>), grid, dim3(256), 0, s, __VA_ARGS__);                      \
        }                                                                                                         \
    } while (0)

// every launcher: 0 = launched, > 0 = error, -1 = not taken (the caller runs the generic kernel)
int ew_stream_bn_apply(int dtype, const void* z, int64_t M, int C, const float* scale, const float* shift, int act, void* out,
                       hipStream_t s) {
    if (!stream_ok(dtype, M, C) || !act3(act)) return -1;
    const int64_t nvec = M * (C / 8);
    const bool nt = stream_nt(nvec);
    EW_ACT_NT(bn_apply_stream_k, act, nt, dim3(sweep_grid(nvec)), s, (const u32x4*)z, nvec, C / 8, scale, shift, (u32x4*)out);
    PAI_LAUNCH_CHECK();
    return 0;
}

template <int ACT_A, int ACT>
static void launch_bn2(bool affb, bool nt, int grid, hipStream_t s, const u32x4* za, const float* sca, const float* sha,
                       const u32x4* zb, const float* scb, const float* shb, int64_t nvec, int G, u32x4* out) {
    if (affb) {
        if (nt) PAI_LAUNCH((bn2_add_act_stream_k<ACT_A, ACT, true, true>), dim3(grid), dim3(256), 0, s, za, sca, sha, zb, scb, shb, nvec, G, out);
        else PAI_LAUNCH((bn2_add_act_stream_k<ACT_A, ACT, true, false>), dim3(grid), dim3(256), 0, s, za, sca, sha, zb, scb, shb, nvec, G, out);
    } else {
        if (nt) PAI_LAUNCH((bn2_add_act_stream_k<ACT_A, ACT, false, true>), dim3(grid), dim3(256), 0, s, za, sca, sha, zb, scb, shb, nvec, G, out);
        else PAI_LAUNCH((bn2_add_act_stream_k<ACT_A, ACT, false, false>), dim3(grid), dim3(256), 0, s, za, sca, sha, zb, scb, shb, nvec, G, out);
    }
}

int ew_stream_bn2_add_act(int dtype, const void* za, const float* sca, const float* sha, const void* zb, const float* scb,
                          const float* shb, int64_t M, int C, int act_a, int act, void* out, hipStream_t s) {
    if (!stream_ok(dtype, M, C)) return -1;
    const bool ra = act_a == PAI_ACT_RELU, r = act == PAI_ACT_RELU;
    if ((!ra && act_a != PAI_ACT_NONE) || (!r && act != PAI_ACT_NONE)) return -1;      // LeakyReLU tails: generic kernel
    const int64_t nvec = M * (C / 8);
    const bool nt = stream_nt(nvec), affb = scb != nullptr;
    const int grid = sweep_grid(nvec), G = C / 8;
    const u32x4 *a = (const u32x4*)za, *b = (const u32x4*)zb;
    u32x4* o = (u32x4*)out;
    if (ra && r) launch_bn2<PAI_ACT_RELU, PAI_ACT_RELU>(affb, nt, grid, s, a, sca, sha, b, scb, shb, nvec, G, o);
    else if (ra) launch_bn2<PAI_ACT_RELU, PAI_ACT_NONE>(affb, nt, grid, s, a, sca, sha, b, scb, shb, nvec, G, o);
    else if (r) launch_bn2<PAI_ACT_NONE, PAI_ACT_RELU>(affb, nt, grid, s, a, sca, sha, b, scb, shb, nvec, G, o);
    else launch_bn2<PAI_ACT_NONE, PAI_ACT_NONE>(affb, nt, grid, s, a, sca, sha, b, scb, shb, nvec, G, o);
    PAI_LAUNCH_CHECK();
    return 0;
}

int ew_stream_add_act(int dtype, const void* a, const void* b, int64_t numel, int act, void* out, hipStream_t s) {
    // no channel structure: "rows" of 8 elements
    if (dtype != PAI_BF16 || numel % 8 != 0 || numel < (1 << 18) || !act3(act) || !pai_tunable("ew_stream", 1)) return -1;
    const int64_t nvec = numel / 8;
    const bool nt = stream_nt(nvec);
    EW_ACT_NT(add_act_stream_k, act, nt, dim3(sweep_grid(nvec)), s, (const u32x4*)a, (const u32x4*)b, nvec, (u32x4*)out);
    PAI_LAUNCH_CHECK();
    return 0;
}

// pass 1 without a stored du, a second gradient or a stored activation (what the composable networks call)
int ew_stream_bn_bwd_reduce(int dtype, const void* g, int act, const void* z, int64_t M, int C, const float* scale,
                            const float* shift, const float* mean, const float* rstd, float* partials, int rows,
                            hipStream_t s) {
    if (!stream_ok(dtype, M, C) || !act3(act) || (act != PAI_ACT_NONE && !(scale && shift))) return -1;
    const int64_t rpb = (M + rows - 1) / rows;
    const bool nt = stream_nt(M * (C / 8));
    EW_ACT_NT(bn_bwd_reduce_stream_k, act, nt, dim3(rows), s, (const u32x4*)g, (const u32x4*)z, M, C / 8, rpb, mean, rstd, scale,
              shift, partials);
    PAI_LAUNCH_CHECK();
    return 0;
}

int ew_stream_bn_bwd_apply(int dtype, const void* g, int act, const void* z, int64_t M, int C, const float* scale,
                           const float* shift, const float* mean, const float* rstd, const float* gamma, const float* sums,
                           void* dz, hipStream_t s) {
    if (!stream_ok(dtype, M, C) || !act3(act) || (act != PAI_ACT_NONE && !(scale && shift))) return -1;
    const int64_t nvec = M * (C / 8);
    const bool nt = stream_nt(nvec);
    const float inv_m = (float)(1.0 / (double)M);
    EW_ACT_NT(bn_bwd_apply_stream_k, act, nt, dim3(sweep_grid(nvec)), s, (const u32x4*)g, (const u32x4*)z, nvec, C / 8, inv_m, mean,
              rstd, gamma, sums, scale, shift, (u32x4*)dz);
    PAI_LAUNCH_CHECK();
    return 0;
}

// both BatchNorms of a residual block's tail (see the kernels): -1 when the shape keeps the generic one-branch kernels
int ew_stream_bn2_bwd_reduce(int dtype, const void* d, int act_a, const void* za, const void* zb, int64_t M, int C,
                             const float* scale_a, const float* shift_a, const float* mean_a, const float* rstd_a,
                             const float* mean_b, const float* rstd_b, float* part_a, float* part_b, int rows, hipStream_t s) {
    if (!stream_ok(dtype, M, C) || !act3(act_a) || (act_a != PAI_ACT_NONE && !(scale_a && shift_a))) return -1;
    const int64_t rpb = (M + rows - 1) / rows;
    const bool nt = stream_nt(M * (C / 8));
    EW_ACT_NT(bn2_bwd_reduce_stream_k, act_a, nt, dim3(rows), s, (const u32x4*)d, (const u32x4*)za, (const u32x4*)zb, M, C / 8, rpb,
              mean_a, rstd_a, scale_a, shift_a, mean_b, rstd_b, part_a, part_b);
    PAI_LAUNCH_CHECK();
    return 0;
}

int ew_stream_bn2_bwd_apply(int dtype, const void* d, int act_a, const void* za, const void* zb, int64_t M, int C,
                            const float* scale_a, const float* shift_a, const float* mean_a, const float* rstd_a,
                            const float* gamma_a, const float* sums_a, const float* mean_b, const float* rstd_b,
                            const float* gamma_b, const float* sums_b, void* dza, void* dzb, hipStream_t s) {
    if (!stream_ok(dtype, M, C) || !act3(act_a) || (act_a != PAI_ACT_NONE && !(scale_a && shift_a))) return -1;
    const int64_t nvec = M * (C / 8);
    const bool nt = stream_nt(nvec);
    const float inv_m = (float)(1.0 / (double)M);
    EW_ACT_NT(bn2_bwd_apply_stream_k, act_a, nt, dim3(sweep_grid(nvec)), s, (const u32x4*)d, (const u32x4*)za, (const u32x4*)zb, nvec,
              C / 8, inv_m, mean_a, rstd_a, gamma_a, sums_a, scale_a, shift_a, mean_b, rstd_b, gamma_b, sums_b, (u32x4*)dza,
              (u32x4*)dzb);
    PAI_LAUNCH_CHECK();
    return 0;
}

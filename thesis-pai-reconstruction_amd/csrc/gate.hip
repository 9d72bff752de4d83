// Attention gate of the Attention U-Net skip connections (reference models/attention_unet.py:48-96):
//   h   = ReLU(BN_s(conv1x1_s(signal)) + BN_i(conv1x1_i(x)))          C -> K = C/2 channels
//   att = Sigmoid(BN_a(conv1x1_a(h)))                                  K -> 1
//   out = x * att
// The two C -> K pointwise convolutions (and their gradients) run through the gather-GEMM kernels
// (pai_conv_* with kernel = 1); BatchNorm statistics / finalisation through the bn.hip entry points.
// This file holds what is left, all HBM-bound row kernels over NHWC tensors [M][C]:
//   gate_hidden_k      h, logit = <h, w_a> + b_a, and the (sum, sum^2) partials of logit for BN_a
//   gate_apply_k       att = sigmoid(logit * sc + sh), out = x * att
//   gate_apply_bwd_k   dx_skip = dout * att, d att = <dout, x>, through the sigmoid, BN_a partial sums
//   gate_hidden_bwd_k  BN_a backward, d h through w_a and ReLU, d w_a / d b_a, BN_i / BN_s partial sums
// thread = 8 consecutive channels of one row; a row is covered by C/8 (K/8) neighbouring lanes.
#include "common.h"

constexpr int GATE_MAX_BLOCKS = 2048;

static int gate_blocks(int64_t M) {
    int64_t b = (M + 63) / 64;
    if (b > GATE_MAX_BLOCKS) b = GATE_MAX_BLOCKS;
    return b < 1 ? 1 : (int)b;
}
extern "C" int pai_gate_partial_rows(int64_t M) { return gate_blocks(M); }

// sum over the `span` lanes (power of two <= 64) that share a row
__device__ __forceinline__ float span_sum(float v, int span) {
    for (int o = 1; o < span; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- forward --------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gate_hidden_k(const T* ig, const T* sg, int64_t M, int K,
                                                     const float* sc_i, const float* sh_i, const float* sc_s,
                                                     const float* sh_s, const float* wa, const float* ba,
                                                     T* h, float* logit, float* partials, int64_t rows_per_block) {
    __shared__ float red[2][256];
    const int span = K / 8;                  // lanes per row (4 .. 64)
    const int tid = threadIdx.x, kc = (tid % span) * 8, rl = tid / span, lanes = 256 / span;
    float si[8], hi[8], ss[8], hs[8], w[8];
    V8<float>::ld(sc_i + kc, si); V8<float>::ld(sh_i + kc, hi);
    V8<float>::ld(sc_s + kc, ss); V8<float>::ld(sh_s + kc, hs);
    V8<float>::ld(wa + kc, w);
    const float bias = ba ? ba[0] : 0.f;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float s = 0.f, q = 0.f;
    for (int64_t rb = r0; rb < r1; rb += lanes) {     // uniform trip count: shuffles need every lane
        const int64_t r = rb + rl;
        const bool valid = r < r1;
        float dot = 0.f;
        if (valid) {
            float a[8], b[8], hv[8];
            V8<T>::ld(ig + r * K + kc, a);
            V8<T>::ld(sg + r * K + kc, b);
#pragma unroll
            for (int k = 0; k < 8; ++k) hv[k] = fmaxf(fmaf(a[k], si[k], hi[k]) + fmaf(b[k], ss[k], hs[k]), 0.f);
            V8<T>::st(h + r * K + kc, hv);
            if (sizeof(T) == 2) {   // the logit is formed from h as stored (what the backward pass re-reads)
#pragma unroll
                for (int k = 0; k < 8; ++k) hv[k] = bf2f(f2bf(hv[k]));
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) dot = fmaf(hv[k], w[k], dot);
        }
        dot = span_sum(dot, span);
        if (valid && (tid % span) == 0) {
            const float l = dot + bias;
            logit[r] = l;
            s += l;
            q = fmaf(l, l, q);
        }
    }
    red[0][tid] = s;
    red[1][tid] = q;
    __syncthreads();
    if (tid < 2) {
        float t = 0.f;
        for (int i = 0; i < 256; ++i) t += red[tid][i];
        partials[(size_t)blockIdx.x * 2 + tid] = t;     // [block][2][C = 1]
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gate_apply_k(const T* x, const float* logit, int64_t M, int C,
                                                    const float* sc_a, const float* sh_a, T* out, float* att) {
    const int span = C / 8;
    const float sc = sc_a[0], sh = sh_a[0];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < M * span; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / span;
        const float a = 1.f / (1.f + __expf(-fmaf(logit[r], sc, sh)));
        float v[8];
        V8<T>::ld(x + i * 8, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] *= a;
        V8<T>::st(out + i * 8, v);
        if (i - r * span == 0) att[r] = a;
    }
}

// ---- backward -------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gate_apply_bwd_k(const T* dout, const T* x, const float* att,
                                                        const float* logit, int64_t M, int C, const float* mean_a,
                                                        const float* rstd_a, T* dx_skip, float* dl, float* partials,
                                                        int64_t rows_per_block, int relu_out) {
    __shared__ float red[2][256];
    const int tid = threadIdx.x;
    const float mu = mean_a[0], rs = rstd_a[0];
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float s1 = 0.f, s2 = 0.f;
    // C/8 lanes per row; C = 512 -> a whole wave per row, C = 1024 would need two passes (not in the reference)
    const int span = C / 8 > 64 ? 64 : C / 8, lanes = 256 / span;
    const int cl = tid % span, rl = tid / span;
    for (int64_t rb = r0; rb < r1; rb += lanes) {
        const int64_t r = rb + rl;
        const bool valid = r < r1;
        float dot = 0.f;
        const float a = valid ? att[r] : 0.f;
        if (valid) {
            for (int c = cl * 8; c < C; c += span * 8) {
                float d[8], xv[8], o[8];
                V8<T>::ld(dout + r * C + c, d);
                V8<T>::ld(x + r * C + c, xv);
                if (relu_out) {   // the consumer read ReLU(out); att > 0, so sign(out) = sign(x)
#pragma unroll
                    for (int k = 0; k < 8; ++k) d[k] = xv[k] > 0.f ? d[k] : 0.f;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) { dot = fmaf(d[k], xv[k], dot); o[k] = d[k] * a; }
                V8<T>::st(dx_skip + r * C + c, o);
            }
        }
        dot = span_sum(dot, span);
        if (valid && cl == 0) {
            const float g = dot * a * (1.f - a);       // through the sigmoid: gradient w.r.t. BN_a's output
            dl[r] = g;
            s1 += g;
            s2 = fmaf(g, (logit[r] - mu) * rs, s2);
        }
    }
    red[0][tid] = s1;
    red[1][tid] = s2;
    __syncthreads();
    if (tid < 2) {
        float t = 0.f;
        for (int i = 0; i < 256; ++i) t += red[tid][i];
        partials[(size_t)blockIdx.x * 2 + tid] = t;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gate_hidden_bwd_k(const float* dl, const float* logit, const T* h, const T* ig,
                                                         const T* sg, int64_t M, int K, const float* mean_a,
                                                         const float* rstd_a, const float* gamma_a, const float* sums_a,
                                                         const float* wa, const float* mean_i, const float* rstd_i,
                                                         const float* mean_s, const float* rstd_s, T* dsum,
                                                         float* part_i, float* part_s, float* dwa, float* dba,
                                                         int64_t rows_per_block) {
    __shared__ float red[256][9];
    const int span = K / 8;
    const int tid = threadIdx.x, kc = (tid % span) * 8, rl = tid / span, lanes = 256 / span;
    const float mu = mean_a[0], rs = rstd_a[0], ga = gamma_a ? gamma_a[0] : 1.f;
    const float inv_m = (float)(1.0 / (double)M);
    const float m0 = sums_a[0] * inv_m, m1 = sums_a[1] * inv_m;
    float w[8], mi[8], ri[8], ms[8], rsv[8];
    V8<float>::ld(wa + kc, w);
    V8<float>::ld(mean_i + kc, mi); V8<float>::ld(rstd_i + kc, ri);
    V8<float>::ld(mean_s + kc, ms); V8<float>::ld(rstd_s + kc, rsv);
    float a0[8], a1[8], a2[8], a3[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a0[k] = a1[k] = a2[k] = a3[k] = 0.f;
    float db = 0.f;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    for (int64_t r = r0 + rl; r < r1; r += lanes) {
        // BatchNorm(1) backward (reference attention_unet.py:83): d logit from the gradient of its output
        const float xh = (logit[r] - mu) * rs;
        const float dlog = ga * rs * (dl[r] - m0 - xh * m1);
        float hv[8], iv[8], sv[8], d[8];
        V8<T>::ld(h + r * K + kc, hv);
        V8<T>::ld(ig + r * K + kc, iv);
        V8<T>::ld(sg + r * K + kc, sv);
#pragma unroll
        for (int k = 0; k < 8; ++k) d[k] = hv[k] > 0.f ? dlog * w[k] : 0.f;
        V8<T>::st(dsum + r * K + kc, d);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float dd = d[k];
            if (sizeof(T) == 2) dd = bf2f(f2bf(dd));    // statistics from the value as stored
            a0[k] += dd;
            a1[k] = fmaf(dd, (iv[k] - mi[k]) * ri[k], a1[k]);
            a2[k] = fmaf(dd, (sv[k] - ms[k]) * rsv[k], a2[k]);
            a3[k] = fmaf(dlog, hv[k], a3[k]);
        }
        if ((tid % span) == 0) db += dlog;
    }
    // combine the row lanes of the block, one quantity at a time: thread c < K sums channel c over the `lanes`
    // row lanes (a first version let K/8 threads walk all lanes x 8 channels serially: 2048 dependent LDS reads,
    // most of this kernel's time on the 1 M-pixel level)
#pragma unroll
    for (int qn = 0; qn < 4; ++qn) {
        const float* src = qn == 0 ? a0 : qn == 1 ? a1 : qn == 2 ? a2 : a3;
#pragma unroll
        for (int k = 0; k < 8; ++k) red[tid][k] = src[k];
        red[tid][8] = db;
        __syncthreads();
        for (int c = tid; c < K; c += 256) {
            const int sp = c >> 3, k = c & 7;
            float t = 0.f;
            for (int l = 0; l < lanes; ++l) t += red[sp + l * span][k];
            const size_t row = (size_t)blockIdx.x * 2;
            if (qn == 0) { part_i[(row + 0) * K + c] = t; part_s[(row + 0) * K + c] = t; }
            else if (qn == 1) part_i[(row + 1) * K + c] = t;
            else if (qn == 2) part_s[(row + 1) * K + c] = t;
            else atomicAdd(dwa + c, t);
        }
        if (qn == 3 && tid == 255 && dba) {
            float t = 0.f;
            for (int l = 0; l < lanes; ++l) t += red[l * span][8];
            atomicAdd(dba, t);
        }
        __syncthreads();
    }
}

// ---- entry points ---------------------------------------------------------------------------
static int check_k(int K, const char* what) {
    PAI_CHECK(K >= 8 && K <= 512 && (K & (K - 1)) == 0, "%s: channel count %d must be a power of two in [8, 512]", what, K);
    return 0;
}

extern "C" int pai_gate_hidden(int dtype, const void* ig, const void* sg, int64_t M, int K, const float* scale_i,
                               const float* shift_i, const float* scale_s, const float* shift_s, const float* w_a,
                               const float* b_a, void* h, float* logit, float* partials, void* stream) {
    PAI_CHECK(ig && sg && scale_i && shift_i && scale_s && shift_s && w_a && h && logit && partials,
              "pai_gate_hidden: null pointer");
    if (check_k(K, "pai_gate_hidden")) return 1;
    const int blocks = gate_blocks(M);
    const int64_t rpb = (M + blocks - 1) / blocks;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PAI_F32)
        PAI_LAUNCH(gate_hidden_k<float>, dim3(blocks), dim3(256), 0, s, (const float*)ig, (const float*)sg, M, K,
                           scale_i, shift_i, scale_s, shift_s, w_a, b_a, (float*)h, logit, partials, rpb);
    else
        PAI_LAUNCH(gate_hidden_k<bf16_t>, dim3(blocks), dim3(256), 0, s, (const bf16_t*)ig, (const bf16_t*)sg, M,
                           K, scale_i, shift_i, scale_s, shift_s, w_a, b_a, (bf16_t*)h, logit, partials, rpb);
    PAI_LAUNCH_CHECK();
    return 0;
}

extern "C" int pai_gate_apply(int dtype, const void* x, const float* logit, int64_t M, int C, const float* scale_a,
                              const float* shift_a, void* out, float* att, void* stream) {
    PAI_CHECK(x && logit && scale_a && shift_a && out && att, "pai_gate_apply: null pointer");
    PAI_CHECK(C % 8 == 0, "pai_gate_apply: C=%d must be a multiple of 8", C);
    int64_t b = (M * (C / 8) + 255) / 256;
    if (b > 8192) b = 8192;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PAI_F32)
        PAI_LAUNCH(gate_apply_k<float>, dim3((int)b), dim3(256), 0, s, (const float*)x, logit, M, C, scale_a,
                           shift_a, (float*)out, att);
    else
        PAI_LAUNCH(gate_apply_k<bf16_t>, dim3((int)b), dim3(256), 0, s, (const bf16_t*)x, logit, M, C, scale_a,
                           shift_a, (bf16_t*)out, att);
    PAI_LAUNCH_CHECK();
    return 0;
}

extern "C" int pai_gate_apply_bwd(int dtype, const void* dout, const void* x, const float* att, const float* logit,
                                  int64_t M, int C, const float* mean_a, const float* rstd_a, void* dx_skip, float* dl,
                                  float* partials, int relu_out, void* stream) {
    PAI_CHECK(dout && x && att && logit && mean_a && rstd_a && dx_skip && dl && partials, "pai_gate_apply_bwd: null pointer");
    PAI_CHECK(C >= 8 && (C & (C - 1)) == 0, "pai_gate_apply_bwd: C=%d must be a power of two >= 8", C);
    const int blocks = gate_blocks(M);
    const int64_t rpb = (M + blocks - 1) / blocks;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PAI_F32)
        PAI_LAUNCH(gate_apply_bwd_k<float>, dim3(blocks), dim3(256), 0, s, (const float*)dout, (const float*)x,
                           att, logit, M, C, mean_a, rstd_a, (float*)dx_skip, dl, partials, rpb, relu_out);
    else
        PAI_LAUNCH(gate_apply_bwd_k<bf16_t>, dim3(blocks), dim3(256), 0, s, (const bf16_t*)dout,
                           (const bf16_t*)x, att, logit, M, C, mean_a, rstd_a, (bf16_t*)dx_skip, dl, partials, rpb, relu_out);
    PAI_LAUNCH_CHECK();
    return 0;
}

extern "C" int pai_gate_hidden_bwd(int dtype, const float* dl, const float* logit, const void* h, const void* ig,
                                   const void* sg, int64_t M, int K, const float* mean_a, const float* rstd_a,
                                   const float* gamma_a, const float* sums_a, const float* w_a, const float* mean_i,
                                   const float* rstd_i, const float* mean_s, const float* rstd_s, void* dsum,
                                   float* partials_i, float* partials_s, float* dw_a, float* db_a, void* stream) {
    PAI_CHECK(dl && logit && h && ig && sg && mean_a && rstd_a && sums_a && w_a && mean_i && rstd_i && mean_s &&
                  rstd_s && dsum && partials_i && partials_s && dw_a,
              "pai_gate_hidden_bwd: null pointer");
    if (check_k(K, "pai_gate_hidden_bwd")) return 1;
    const int blocks = gate_blocks(M);
    const int64_t rpb = (M + blocks - 1) / blocks;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PAI_F32)
        PAI_LAUNCH(gate_hidden_bwd_k<float>, dim3(blocks), dim3(256), 0, s, dl, logit, (const float*)h,
                           (const float*)ig, (const float*)sg, M, K, mean_a, rstd_a, gamma_a, sums_a, w_a, mean_i, rstd_i,
                           mean_s, rstd_s, (float*)dsum, partials_i, partials_s, dw_a, db_a, rpb);
    else
        PAI_LAUNCH(gate_hidden_bwd_k<bf16_t>, dim3(blocks), dim3(256), 0, s, dl, logit, (const bf16_t*)h,
                           (const bf16_t*)ig, (const bf16_t*)sg, M, K, mean_a, rstd_a, gamma_a, sums_a, w_a, mean_i,
                           rstd_i, mean_s, rstd_s, (bf16_t*)dsum, partials_i, partials_s, dw_a, db_a, rpb);
    PAI_LAUNCH_CHECK();
    return 0;
}

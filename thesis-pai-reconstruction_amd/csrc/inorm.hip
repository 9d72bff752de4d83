// nn.InstanceNorm2d(C) (affine=False, no running statistics, eps 1e-5) with the LeakyReLU / ReLU behind it, forward and
// backward, over NHWC tensors: the optional normalisation of DiscriminatorBlock (reference models/wrapper.py:176-209,
// `norm=True`; the reference's own Discriminator never enables it).
//   forward   y = act((x - mean[n][c]) * rstd[n][c]),  mean / biased variance over the H x W pixels of sample n
//   backward  du = g * act'(y);  dx = rstd * (du - mean(du) - xhat * mean(du * xhat)),  xhat rebuilt from x
// One workgroup = one sample x 64 channels: 8 chunk lanes x 32 pixel lanes sweep the sample twice (statistics, then the
// normalisation; the second sweep comes from L2 for the sizes a discriminator sees).  Sums in fp32 per lane (<= HW / 32
// terms), fp64 across the lanes.
#include "common.h"

namespace {
constexpr int IN_CH = 64;        // channels per workgroup

__device__ __forceinline__ void in_block_sums(double (*red)[32][64], const float* s1, const float* s2, int pl, int cl,
                                              float* o1, float* o2) {
    // red[k][pixel lane][channel]: this thread owns channels cl * 8 .. + 8
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[0][pl][cl * 8 + e] = (double)s1[e]; red[1][pl][cl * 8 + e] = (double)s2[e]; }
    __syncthreads();
    const int t = threadIdx.x;
    if (t < 128) {
        const int k = t >> 6, c = t & 63;
        double a = 0.0;
        for (int l = 0; l < 32; ++l) a += red[k][l][c];
        red[k][0][c] = a;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) { o1[e] = (float)red[0][0][cl * 8 + e]; o2[e] = (float)red[1][0][cl * 8 + e]; }
}
}  // namespace

template <typename T>
__global__ __launch_bounds__(256) void instnorm_fwd_k(const T* x, int HW, int C, float eps, int act, T* y, float* mean_o,
                                                      float* rstd_o) {
    __shared__ double red[2][32][64];
    const int n = blockIdx.y, c0 = blockIdx.x * IN_CH;
    const int cl = threadIdx.x & 7, pl = threadIdx.x >> 3;
    const bool cv = c0 + cl * 8 < C;
    const T* xs = x + (size_t)n * HW * C + c0 + cl * 8;
    T* ys = y + (size_t)n * HW * C + c0 + cl * 8;
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
    if (cv)
        for (int p = pl; p < HW; p += 32) {
            float v[8];
            V8<T>::ld(xs + (size_t)p * C, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) { s1[e] += v[e]; s2[e] = fmaf(v[e], v[e], s2[e]); }
        }
    float t1[8], t2[8];
    in_block_sums(red, s1, s2, pl, cl, t1, t2);
    if (!cv) return;
    float mu[8], rs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const double m = (double)t1[e] / HW;
        double var = (double)t2[e] / HW - m * m;
        if (var < 0.0) var = 0.0;
        mu[e] = (float)m;
        rs[e] = (float)(1.0 / sqrt(var + (double)eps));
    }
    if (pl == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            mean_o[(size_t)n * C + c0 + cl * 8 + e] = mu[e];
            rstd_o[(size_t)n * C + c0 + cl * 8 + e] = rs[e];
        }
    }
    for (int p = pl; p < HW; p += 32) {
        float v[8];
        V8<T>::ld(xs + (size_t)p * C, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = act_apply((v[e] - mu[e]) * rs[e], act);
        V8<T>::st(ys + (size_t)p * C, v);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void instnorm_bwd_k(const T* g, const T* x, int HW, int C, int act, const float* mean,
                                                      const float* rstd, T* dx) {
    __shared__ double red[2][32][64];
    const int n = blockIdx.y, c0 = blockIdx.x * IN_CH;
    const int cl = threadIdx.x & 7, pl = threadIdx.x >> 3;
    const bool cv = c0 + cl * 8 < C;
    const size_t base = (size_t)n * HW * C + c0 + cl * 8;
    float mu[8], rs[8], s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        mu[e] = cv ? mean[(size_t)n * C + c0 + cl * 8 + e] : 0.f;
        rs[e] = cv ? rstd[(size_t)n * C + c0 + cl * 8 + e] : 0.f;
        s1[e] = s2[e] = 0.f;
    }
    if (cv)
        for (int p = pl; p < HW; p += 32) {
            float gv[8], xv[8];
            V8<T>::ld(g + base + (size_t)p * C, gv);
            V8<T>::ld(x + base + (size_t)p * C, xv);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xh = (xv[e] - mu[e]) * rs[e];
                const float du = gv[e] * act_grad(xh, act);       // sign(act(xhat)) = sign(xhat)
                s1[e] += du;
                s2[e] = fmaf(du, xh, s2[e]);
            }
        }
    float t1[8], t2[8];
    in_block_sums(red, s1, s2, pl, cl, t1, t2);
    if (!cv) return;
    const float inv = 1.f / (float)HW;
    for (int p = pl; p < HW; p += 32) {
        float gv[8], xv[8];
        V8<T>::ld(g + base + (size_t)p * C, gv);
        V8<T>::ld(x + base + (size_t)p * C, xv);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float xh = (xv[e] - mu[e]) * rs[e];
            const float du = gv[e] * act_grad(xh, act);
            gv[e] = rs[e] * (du - t1[e] * inv - xh * t2[e] * inv);
        }
        V8<T>::st(dx + base + (size_t)p * C, gv);
    }
}

extern "C" int pai_instnorm_fwd(int dtype, const void* x, int N, int HW, int C, float eps, int act, void* y, float* mean,
                                float* rstd, void* stream) {
    PAI_CHECK(x && y && mean && rstd, "pai_instnorm_fwd: null pointer");
    PAI_CHECK(N > 0 && HW > 0 && C > 0 && C % 8 == 0, "pai_instnorm_fwd: N=%d HW=%d C=%d (C must be a multiple of 8)", N, HW, C);
    PAI_CHECK(act == PAI_ACT_NONE || act == PAI_ACT_RELU || act == PAI_ACT_LRELU, "pai_instnorm_fwd: act=%d", act);
    const dim3 grid(cdiv(C, IN_CH), N);
    if (dtype == PAI_F32)
        PAI_LAUNCH(instnorm_fwd_k<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, HW, C, eps, act, (float*)y,
                   mean, rstd);
    else
        PAI_LAUNCH(instnorm_fwd_k<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, HW, C, eps, act,
                   (bf16_t*)y, mean, rstd);
    PAI_LAUNCH_CHECK();
    return 0;
}

extern "C" int pai_instnorm_bwd(int dtype, const void* g, const void* x, int N, int HW, int C, int act, const float* mean,
                                const float* rstd, void* dx, void* stream) {
    PAI_CHECK(g && x && dx && mean && rstd, "pai_instnorm_bwd: null pointer");
    PAI_CHECK(N > 0 && HW > 0 && C > 0 && C % 8 == 0, "pai_instnorm_bwd: N=%d HW=%d C=%d (C must be a multiple of 8)", N, HW, C);
    PAI_CHECK(act == PAI_ACT_NONE || act == PAI_ACT_RELU || act == PAI_ACT_LRELU, "pai_instnorm_bwd: act=%d", act);
    const dim3 grid(cdiv(C, IN_CH), N);
    if (dtype == PAI_F32)
        PAI_LAUNCH(instnorm_bwd_k<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)g, (const float*)x, HW, C, act,
                   mean, rstd, (float*)dx);
    else
        PAI_LAUNCH(instnorm_bwd_k<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)g, (const bf16_t*)x, HW, C, act,
                   mean, rstd, (bf16_t*)dx);
    PAI_LAUNCH_CHECK();
    return 0;
}

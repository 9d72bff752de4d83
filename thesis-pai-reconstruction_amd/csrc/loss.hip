// Mean-reduced losses with fused gradients (HBM-bound, one pass).
// Replaces F.binary_cross_entropy_with_logits / F.l1_loss / F.mse_loss at
// models/wrapper.py:45-49,66,84-93 and the tanh backward of models/pix2pix.py:216.
#include "common.h"

__device__ __forceinline__ void block_atomic_add(double v, double* dst) {
    __shared__ double wsum[4];
    // wave reduce in double
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) wsum[wid] = v;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(dst, wsum[0] + wsum[1] + wsum[2] + wsum[3]);
}

enum { L_BCE = 0, L_L1 = 1, L_MSE = 2 };

template <int KIND>
__global__ __launch_bounds__(256) void loss_k(const float* x, const float* t, float tconst, int64_t numel,
                                              double loss_scale_over_n, double* loss, float gscale,
                                              float* grad) {
    double acc = 0.0;
    // L1 / MSE over 16-byte vectors, four per stream in flight per thread, 512 workgroups (launch_loss): every workgroup ends in
    // ONE fp64 atomic on the same address, ~15 ns each -- with 2048 workgroups of one element per thread and iteration the
    // generator's L1 term (2 x 16.8 MB read, 16.8 MB written) took 34 us on the critical path between the discriminator's
    // input gradient and the generator's backward pass, most of it that queue.  Same per-element arithmetic, fp64 accumulation.
    if (KIND != L_BCE && (numel & 3) == 0 && ((((uintptr_t)x | (uintptr_t)t | (uintptr_t)grad) & 15) == 0)) {
        const int64_t n4 = numel >> 2, stride = (int64_t)gridDim.x * 256;
        for (int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x; i0 < n4; i0 += 4 * stride) {
            float4 xv[4], tv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t i = i0 + u * stride;
                xv[u] = i < n4 ? ((const float4*)x)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
                tv[u] = i < n4 ? ((const float4*)t)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t i = i0 + u * stride;
                if (i >= n4) break;
                const float xs[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w}, ts[4] = {tv[u].x, tv[u].y, tv[u].z, tv[u].w};
                float gs[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = xs[e] - ts[e];
                    acc += (double)(KIND == L_L1 ? fabsf(d) : d * d);
                    gs[e] = (KIND == L_L1 ? (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) : 2.f * d) * gscale;
                }
                if (grad) ((float4*)grad)[i] = make_float4(gs[0], gs[1], gs[2], gs[3]);
            }
        }
        block_atomic_add(acc * loss_scale_over_n, loss);
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < numel; i += (int64_t)gridDim.x * 256) {
        const float xv = x[i];
        float l, g;
        if (KIND == L_BCE) {
            // max(x,0) - x*t + log1p(exp(-|x|));  d/dx = sigmoid(x) - t
            l = fmaxf(xv, 0.f) - xv * tconst + log1pf(expf(-fabsf(xv)));
            g = 1.f / (1.f + expf(-xv)) - tconst;
        } else if (KIND == L_L1) {
            const float d = xv - t[i];
            l = fabsf(d);
            g = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        } else {
            const float d = xv - t[i];
            l = d * d;
            g = 2.f * d;
        }
        acc += (double)l;
        if (grad) grad[i] = g * gscale;
    }
    block_atomic_add(acc * loss_scale_over_n, loss);
}

template <int KIND>
static int launch_loss(const float* x, const float* t, float tconst, int64_t numel, float loss_scale,
                       double* loss, float grad_scale, float* grad, void* stream) {
    PAI_CHECK(x && loss && numel > 0, "loss: null pointer / empty");
    int64_t blocks = (numel + 256 * 8 - 1) / (256 * 8);
    if (blocks > 2048) blocks = 2048;
    if (KIND != L_BCE && blocks > 512) blocks = 512;      // one same-address fp64 atomic per workgroup (see the kernel)
    if (blocks < 1) blocks = 1;
    PAI_LAUNCH(loss_k<KIND>, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, x, t, tconst,
                       numel, (double)loss_scale / (double)numel, loss,
                       (float)((double)grad_scale / (double)numel), grad);
    PAI_LAUNCH_CHECK();
    return 0;
}

extern "C" int pai_bce_logits(const float* logits, int64_t numel, float target, float loss_scale,
                              double* loss, float grad_scale, float* grad, void* stream) {
    return launch_loss<L_BCE>(logits, nullptr, target, numel, loss_scale, loss, grad_scale, grad, stream);
}
extern "C" int pai_l1(const float* pred, const float* target, int64_t numel, float loss_scale, double* loss,
                      float grad_scale, float* grad, void* stream) {
    PAI_CHECK(target, "pai_l1: null target");
    return launch_loss<L_L1>(pred, target, 0.f, numel, loss_scale, loss, grad_scale, grad, stream);
}
extern "C" int pai_mse(const float* pred, const float* target, int64_t numel, float loss_scale, double* loss,
                       float grad_scale, float* grad, void* stream) {
    PAI_CHECK(target, "pai_mse: null target");
    return launch_loss<L_MSE>(pred, target, 0.f, numel, loss_scale, loss, grad_scale, grad, stream);
}

// The scalar glue of a loss / metric (fp64 accumulator -> fp32 value, log / sqrt of the logged metrics, re-arming the
// accumulator) in ONE single-thread launch instead of the fill / divide / log / sqrt / neg / cast chain of tensor ops
// (5-9 us of launch latency each, ~25 of them per GAN step).
__global__ void scalar_take_k(double* acc, float* out) {
    out[0] = (float)acc[0];
    acc[0] = 0.0;
}
extern "C" int pai_scalar_take(double* acc, float* out, void* stream) {
    PAI_CHECK(acc && out, "pai_scalar_take: null pointer");
    PAI_LAUNCH(scalar_take_k, dim3(1), dim3(1), 0, (hipStream_t)stream, acc, out);
    PAI_LAUNCH_CHECK();
    return 0;
}
__global__ void metrics_take_k(double* sums, double inv_images, double inv_numel, float* out3) {
    const double mse = sums[1] * inv_numel;
    out3[0] = (float)(sums[0] * inv_images);
    out3[1] = (float)(-log(mse) * (10.0 / 2.302585092994045684));      // PSNR, data range 1
    out3[2] = (float)sqrt(mse);                                        // RMSE
    sums[0] = sums[1] = 0.0;
}
extern "C" int pai_metrics_take(double* sums, int64_t n_images, int64_t numel, float* out3, void* stream) {
    PAI_CHECK(sums && out3 && n_images > 0 && numel > 0, "pai_metrics_take: bad arguments");
    PAI_LAUNCH(metrics_take_k, dim3(1), dim3(1), 0, (hipStream_t)stream, sums, 1.0 / (double)n_images,
                       1.0 / (double)numel, out3);
    PAI_LAUNCH_CHECK();
    return 0;
}

template <typename T>
__global__ __launch_bounds__(256) void tanh_bwd_k(const float* pred, const float* ga, const float* gb,
                                                  int64_t numel, T* dh) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < numel; i += (int64_t)gridDim.x * 256) {
        const float y = pred[i];
        float g = (ga ? ga[i] : 0.f) + (gb ? gb[i] : 0.f);
        Conv<T>::st(dh + i, g * (1.f - y * y));
    }
}

extern "C" int pai_tanh_bwd(int dtype, const float* pred, const float* g_a, const float* g_b, int64_t numel,
                            void* dh, void* stream) {
    PAI_CHECK(pred && dh && (g_a || g_b), "pai_tanh_bwd: null pointer");
    int64_t blocks = (numel + 1023) / 1024;
    if (blocks > 4096) blocks = 4096;
    if (dtype == PAI_F32)
        PAI_LAUNCH(tanh_bwd_k<float>, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, pred, g_a,
                           g_b, numel, (float*)dh);
    else
        PAI_LAUNCH(tanh_bwd_k<bf16_t>, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, pred,
                           g_a, g_b, numel, (bf16_t*)dh);
    PAI_LAUNCH_CHECK();
    return 0;
}

// denormalize (models/utils.py:11): out = clamp(x*0.5+0.5, 0, 1); backward: g*0.5 inside [0,1]
__global__ __launch_bounds__(256) void denorm_k(const float* x, const float* g, int64_t numel, float* out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < numel; i += (int64_t)gridDim.x * 256) {
        const float u = x[i] * 0.5f + 0.5f;
        if (g) out[i] = (u >= 0.f && u <= 1.f) ? 0.5f * g[i] : 0.f;
        else out[i] = fminf(fmaxf(u, 0.f), 1.f);
    }
}

extern "C" int pai_denormalize(const float* x, const float* grad_out_or_null, int64_t numel, float* out,
                               void* stream) {
    PAI_CHECK(x && out, "pai_denormalize: null pointer");
    int64_t blocks = (numel + 1023) / 1024;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    PAI_LAUNCH(denorm_k, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, x, grad_out_or_null,
                       numel, out);
    PAI_LAUNCH_CHECK();
    return 0;
}

// Residual U-Net building blocks that are not convolutions (reference models/res_unet.py):
//   nn.MaxPool2d(2) (:199), nn.Upsample(scale_factor=2) nearest (:231), the residual sum
//   conv_block(x) + conv_skip(x) (:74,105,130,171) with the optional ReLU behind it (:71,102),
//   and the grouped 3x3 convolution of ResidualBlockNeXt (:151-157, groups = 32).
// All HBM-bound NHWC row kernels; thread = 8 consecutive channels.
#include "common.h"

static int ew_blocks(int64_t nvec) {
    int64_t b = (nvec + 255) / 256;
    if (b > 8192) b = 8192;
    return b < 1 ? 1 : (int)b;
}

// ---- MaxPool2d(2): out[n][y][x][c] = max of the 2x2 window; idx (2 bits) remembers the arg-max for backward ----
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_k(const T* x, int N, int H, int W, int C, T* out, unsigned char* idx) {
    const int OH = H / 2, OW = W / 2, cv = C / 8;
    const int64_t nvec = (int64_t)N * OH * OW * cv;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        const int c8 = (int)(i % cv);
        int64_t r = i / cv;
        const int ox = (int)(r % OW); r /= OW;
        const int oy = (int)(r % OH);
        const int n = (int)(r / OH);
        float best[8];
        unsigned char bi[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { best[k] = -INFINITY; bi[k] = 0; }
#pragma unroll
        for (int p = 0; p < 4; ++p) {     // row-major window order: the first maximum wins, as in ATen
            float v[8];
            V8<T>::ld(x + (((int64_t)n * H + 2 * oy + (p >> 1)) * W + 2 * ox + (p & 1)) * C + c8 * 8, v);
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (v[k] > best[k] || v[k] != v[k]) { best[k] = v[k]; bi[k] = (unsigned char)p; }
        }
        V8<T>::st(out + i * 8, best);
        if (idx) {
            uint2 pk;
            pk.x = bi[0] | (bi[1] << 8) | (bi[2] << 16) | ((unsigned)bi[3] << 24);
            pk.y = bi[4] | (bi[5] << 8) | (bi[6] << 16) | ((unsigned)bi[7] << 24);
            *(uint2*)(idx + i * 8) = pk;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool2_bwd_k(const T* dout, const unsigned char* idx, int N, int H, int W, int C,
                                                      T* dx) {
    const int OH = H / 2, OW = W / 2, cv = C / 8;
    const int64_t nvec = (int64_t)N * OH * OW * cv;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        const int c8 = (int)(i % cv);
        int64_t r = i / cv;
        const int ox = (int)(r % OW); r /= OW;
        const int oy = (int)(r % OH);
        const int n = (int)(r / OH);
        float g[8];
        V8<T>::ld(dout + i * 8, g);
        const uint2 pk = *(const uint2*)(idx + i * 8);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const unsigned b = ((k < 4 ? pk.x : pk.y) >> (8 * (k & 3))) & 0xffu;
                v[k] = b == (unsigned)p ? g[k] : 0.f;
            }
            V8<T>::st(dx + (((int64_t)n * H + 2 * oy + (p >> 1)) * W + 2 * ox + (p & 1)) * C + c8 * 8, v);
        }
    }
}

// ---- nearest Upsample(scale_factor=2): forward replicates, backward sums the 2x2 window ----
template <typename T>
__global__ __launch_bounds__(256) void upsample2_k(const T* x, int N, int H, int W, int C, T* out) {
    const int cv = C / 8;
    const int64_t nvec = (int64_t)N * H * W * cv;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        const int c8 = (int)(i % cv);
        int64_t r = i / cv;
        const int ix = (int)(r % W); r /= W;
        const int iy = (int)(r % H);
        const int n = (int)(r / H);
        const uint4 v = *(const uint4*)((const char*)x + i * 8 * sizeof(T));
        const uint4 v2 = sizeof(T) == 4 ? *(const uint4*)((const char*)x + i * 8 * sizeof(T) + 16) : v;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            char* dst = (char*)out + ((((int64_t)n * 2 * H + 2 * iy + (p >> 1)) * 2 * W + 2 * ix + (p & 1)) * C + c8 * 8) * sizeof(T);
            *(uint4*)dst = v;
            if (sizeof(T) == 4) *(uint4*)(dst + 16) = v2;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void upsample2_bwd_k(const T* dout, int N, int H, int W, int C, T* dx) {
    const int cv = C / 8;
    const int64_t nvec = (int64_t)N * H * W * cv;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        const int c8 = (int)(i % cv);
        int64_t r = i / cv;
        const int ix = (int)(r % W); r /= W;
        const int iy = (int)(r % H);
        const int n = (int)(r / H);
        float s[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) s[k] = 0.f;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float v[8];
            V8<T>::ld(dout + ((((int64_t)n * 2 * H + 2 * iy + (p >> 1)) * 2 * W + 2 * ix + (p & 1)) * C + c8 * 8), v);
#pragma unroll
            for (int k = 0; k < 8; ++k) s[k] += v[k];
        }
        V8<T>::st(dx + i * 8, s);
    }
}

// ---- out = act(a + b) ----
template <typename T>
__global__ __launch_bounds__(256) void add_act_k(const T* a, const T* b, int64_t nvec, int act, T* out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        float x[8], y[8];
        V8<T>::ld(a + i * 8, x);
        V8<T>::ld(b + i * 8, y);
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = act_apply(x[k] + y[k], act);
        V8<T>::st(out + i * 8, x);
    }
}

extern "C" int pai_maxpool2(int dtype, const void* x, int N, int H, int W, int C, void* out, unsigned char* idx,
                            void* stream) {
    PAI_CHECK(x && out, "pai_maxpool2: null pointer");
    PAI_CHECK(C % 8 == 0 && H % 2 == 0 && W % 2 == 0, "pai_maxpool2: C=%d must be a multiple of 8, H=%d W=%d even", C, H, W);
    const dim3 grid(ew_blocks((int64_t)N * (H / 2) * (W / 2) * (C / 8)));
    if (dtype == PAI_F32)
        PAI_LAUNCH(maxpool2_k<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, N, H, W, C, (float*)out, idx);
    else
        PAI_LAUNCH(maxpool2_k<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, N, H, W, C, (bf16_t*)out, idx);
    PAI_LAUNCH_CHECK();
    return 0;
}

extern "C" int pai_maxpool2_bwd(int dtype, const void* dout, const unsigned char* idx, int N, int H, int W, int C,
                                void* dx, void* stream) {
    PAI_CHECK(dout && idx && dx, "pai_maxpool2_bwd: null pointer");
    PAI_CHECK(C % 8 == 0 && H % 2 == 0 && W % 2 == 0, "pai_maxpool2_bwd: bad shape");
    const dim3 grid(ew_blocks((int64_t)N * (H / 2) * (W / 2) * (C / 8)));
    if (dtype == PAI_F32)
        PAI_LAUNCH(maxpool2_bwd_k<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)dout, idx, N, H, W, C, (float*)dx);
    else
        PAI_LAUNCH(maxpool2_bwd_k<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dout, idx, N, H, W, C, (bf16_t*)dx);
    PAI_LAUNCH_CHECK();
    return 0;
}

extern "C" int pai_upsample2(int dtype, const void* x, int N, int H, int W, int C, void* out, void* stream) {
    PAI_CHECK(x && out && C % 8 == 0, "pai_upsample2: bad arguments");
    const dim3 grid(ew_blocks((int64_t)N * H * W * (C / 8)));
    if (dtype == PAI_F32)
        PAI_LAUNCH(upsample2_k<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, N, H, W, C, (float*)out);
    else
        PAI_LAUNCH(upsample2_k<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, N, H, W, C, (bf16_t*)out);
    PAI_LAUNCH_CHECK();
    return 0;
}

extern "C" int pai_upsample2_bwd(int dtype, const void* dout, int N, int H, int W, int C, void* dx, void* stream) {
    PAI_CHECK(dout && dx && C % 8 == 0, "pai_upsample2_bwd: bad arguments");
    const dim3 grid(ew_blocks((int64_t)N * H * W * (C / 8)));
    if (dtype == PAI_F32)
        PAI_LAUNCH(upsample2_bwd_k<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)dout, N, H, W, C, (float*)dx);
    else
        PAI_LAUNCH(upsample2_bwd_k<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dout, N, H, W, C, (bf16_t*)dx);
    PAI_LAUNCH_CHECK();
    return 0;
}

extern "C" int pai_add_act(int dtype, const void* a, const void* b, int64_t numel, int act, void* out, void* stream) {
    PAI_CHECK(a && b && out && numel % 8 == 0, "pai_add_act: bad arguments");
    PAI_CHECK(act == PAI_ACT_NONE || act == PAI_ACT_RELU || act == PAI_ACT_LRELU, "pai_add_act: act=%d", act);
    const dim3 grid(ew_blocks(numel / 8));
    if (const int r = ew_stream_add_act(dtype, a, b, numel, act, out, (hipStream_t)stream); r >= 0) return r;
    if (dtype == PAI_F32)
        PAI_LAUNCH(add_act_k<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)a, (const float*)b, numel / 8, act, (float*)out);
    else
        PAI_LAUNCH(add_act_k<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, (const bf16_t*)b, numel / 8, act, (bf16_t*)out);
    PAI_LAUNCH_CHECK();
    return 0;
}

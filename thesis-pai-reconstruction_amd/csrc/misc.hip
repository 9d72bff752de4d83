// Utility kernels: dtype casts, filter re-packing, fp64 row reduction, column sums, fused Adam.
#include "common.h"

template <typename S, typename D>
__global__ __launch_bounds__(256) void cast_k(const S* src, D* dst, int64_t numel) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < numel; i += (int64_t)gridDim.x * 256)
        Conv<D>::st(dst + i, Conv<S>::ld(src + i));
}

extern "C" int pai_cast(int src_dtype, const void* src, int dst_dtype, void* dst, int64_t numel,
                        void* stream) {
    PAI_CHECK(src && dst, "pai_cast: null pointer");
    int64_t blocks = (numel + 1023) / 1024;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipStream_t s = (hipStream_t)stream;
    dim3 g((int)blocks), b(256);
    if (src_dtype == PAI_F32 && dst_dtype == PAI_BF16)
        PAI_LAUNCH((cast_k<float, bf16_t>), g, b, 0, s, (const float*)src, (bf16_t*)dst, numel);
    else if (src_dtype == PAI_BF16 && dst_dtype == PAI_F32)
        PAI_LAUNCH((cast_k<bf16_t, float>), g, b, 0, s, (const bf16_t*)src, (float*)dst, numel);
    else if (src_dtype == PAI_F32 && dst_dtype == PAI_F32)
        PAI_LAUNCH((cast_k<float, float>), g, b, 0, s, (const float*)src, (float*)dst, numel);
    else if (src_dtype == PAI_BF16 && dst_dtype == PAI_BF16)
        PAI_LAUNCH((cast_k<bf16_t, bf16_t>), g, b, 0, s, (const bf16_t*)src, (bf16_t*)dst, numel);
    else
        PAI_CHECK(false, "pai_cast: bad dtypes %d -> %d", src_dtype, dst_dtype);
    PAI_LAUNCH_CHECK();
    return 0;
}

// nn.Dropout2d on an NHWC tensor: out[n][p][c] = x[n][p][c] * mask[n][c]; the gradient is the same product
template <typename T>
__global__ __launch_bounds__(256) void dropout2d_k(const T* x, const float* mask, int64_t nvec, int64_t vec_per_image,
                                                   int C, T* out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        const int64_t n = i / vec_per_image;
        const int c0 = (int)((i * 8) % C);
        float v[8], m[8];
        V8<T>::ld(x + i * 8, v);
        V8<float>::ld(mask + n * C + c0, m);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] *= m[k];
        V8<T>::st(out + i * 8, v);
    }
}

extern "C" int pai_dropout2d(int dtype, const void* x, const float* mask, int N, int64_t HW, int C, void* out,
                             void* stream) {
    PAI_CHECK(x && mask && out, "pai_dropout2d: null pointer");
    PAI_CHECK(C % 8 == 0 && N > 0 && HW > 0, "pai_dropout2d: bad shape N=%d HW=%lld C=%d", N, (long long)HW, C);
    const int64_t vpi = HW * C / 8, nvec = vpi * N;
    int64_t blocks = (nvec + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PAI_F32)
        PAI_LAUNCH(dropout2d_k<float>, dim3((int)blocks), dim3(256), 0, s, (const float*)x, mask, nvec, vpi, C,
                           (float*)out);
    else
        PAI_LAUNCH(dropout2d_k<bf16_t>, dim3((int)blocks), dim3(256), 0, s, (const bf16_t*)x, mask, nvec, vpi,
                           C, (bf16_t*)out);
    PAI_LAUNCH_CHECK();
    return 0;
}

// master [Cout][taps][Cin] fp32 -> fwd pack (same order) and dgrad pack [Cin][taps][Cout]
template <typename D>
__global__ __launch_bounds__(256) void pack_k(const float* w, int Cout, int taps, int Cin, D* wf, D* wd) {
    __shared__ float tile[32][33];
    const int t = blockIdx.z;
    const int co0 = blockIdx.y * 32, ci0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int co = co0 + r, ci = ci0 + tx;
        float v = 0.f;
        if (co < Cout && ci < Cin) {
            const size_t idx = ((size_t)co * taps + t) * Cin + ci;
            v = w[idx];
            if (wf) Conv<D>::st(wf + idx, v);
        }
        tile[r][tx] = v;
    }
    __syncthreads();
    if (wd) {
        for (int r = ty; r < 32; r += 8) {
            const int ci = ci0 + r, co = co0 + tx;
            if (ci < Cin && co < Cout) Conv<D>::st(wd + ((size_t)ci * taps + t) * Cout + co, tile[tx][r]);
        }
    }
}

// bf16 fast path for Cin, Cout multiples of 64: a 64 x 64 tile per workgroup, 16-B loads of the fp32 master and
// 16-B stores of both packs (the 32 x 32 kernel above writes 2 bytes per lane: 2.5 TB/s of the 8 -- 3.2 ms per step
// on the 1.03 B-parameter TransUNet)
__global__ __launch_bounds__(256) void pack64_k(const float* w, int Cout, int taps, int Cin, bf16_t* wf, bf16_t* wd) {
    __shared__ float tile[64][65];
    const int t = blockIdx.z;
    const int co0 = blockIdx.y * 64, ci0 = blockIdx.x * 64;
    const int tid = threadIdx.x;
    // load: row r = tid / 16 + 16 j, float4 column q = tid % 16
    const int q = tid & 15, r0 = tid >> 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = r0 + 16 * j;
        const float4 v = *(const float4*)(w + ((size_t)(co0 + r) * taps + t) * Cin + ci0 + 4 * q);
        tile[r][4 * q + 0] = v.x; tile[r][4 * q + 1] = v.y; tile[r][4 * q + 2] = v.z; tile[r][4 * q + 3] = v.w;
    }
    __syncthreads();
    // store: row = tid / 8 + 32 j, 8-element chunk c = tid % 8
    const int c = tid & 7, s0 = tid >> 3;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = s0 + 32 * j;
        if (wf) {
            unsigned u[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) u[e] = pk2bf(tile[r][8 * c + 2 * e], tile[r][8 * c + 2 * e + 1]);
            *(uint4*)(wf + ((size_t)(co0 + r) * taps + t) * Cin + ci0 + 8 * c) = make_uint4(u[0], u[1], u[2], u[3]);
        }
        if (wd) {   // transposed: row r is an input channel, the chunk 8 output channels
            unsigned u[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) u[e] = pk2bf(tile[8 * c + 2 * e][r], tile[8 * c + 2 * e + 1][r]);
            *(uint4*)(wd + ((size_t)(ci0 + r) * taps + t) * Cout + co0 + 8 * c) = make_uint4(u[0], u[1], u[2], u[3]);
        }
    }
}

// Every 64-multiple layer of a network in ONE launch: after an optimizer step all filter packs are stale at once, and
// 17 launches of 8 us each is launch latency, not work.  Block b belongs to the item whose [blk0, blk0 + blocks) range
// holds it; inside an item the blocks are (input-channel tile, output-channel tile, tap) as in pack64_k.
struct PackItem {
    const float* w;
    bf16_t *wf, *wd;
    int Cout, taps, Cin, blk0;
};
constexpr int PACK_MULTI_MAX = 32;
struct PackMulti {
    PackItem it[PACK_MULTI_MAX];
    int n;
};
__global__ __launch_bounds__(256) void pack64_multi_k(PackMulti pm) {
    __shared__ float tile[64][65];
    int k = 0;
#pragma unroll 1
    for (int i = 1; i < pm.n; ++i)
        if ((int)blockIdx.x >= pm.it[i].blk0) k = i;
    // one item's fields as scalars (the table is indexed by a run-time value)
    const float* w = pm.it[k].w;
    bf16_t *wf = pm.it[k].wf, *wd = pm.it[k].wd;
    const int Cout = pm.it[k].Cout, taps = pm.it[k].taps, Cin = pm.it[k].Cin;
    const int lb = blockIdx.x - pm.it[k].blk0;
    const int nci = Cin / 64, nco = Cout / 64;
    const int ci0 = (lb % nci) * 64, co0 = ((lb / nci) % nco) * 64, t = lb / (nci * nco);
    const int tid = threadIdx.x;
    const int q = tid & 15, r0 = tid >> 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = r0 + 16 * j;
        const float4 v = *(const float4*)(w + ((size_t)(co0 + r) * taps + t) * Cin + ci0 + 4 * q);
        tile[r][4 * q + 0] = v.x; tile[r][4 * q + 1] = v.y; tile[r][4 * q + 2] = v.z; tile[r][4 * q + 3] = v.w;
    }
    __syncthreads();
    const int c = tid & 7, s0 = tid >> 3;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = s0 + 32 * j;
        if (wf) {
            unsigned u[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) u[e] = pk2bf(tile[r][8 * c + 2 * e], tile[r][8 * c + 2 * e + 1]);
            *(uint4*)(wf + ((size_t)(co0 + r) * taps + t) * Cin + ci0 + 8 * c) = make_uint4(u[0], u[1], u[2], u[3]);
        }
        if (wd) {
            unsigned u[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) u[e] = pk2bf(tile[8 * c + 2 * e][r], tile[8 * c + 2 * e + 1][r]);
            *(uint4*)(wd + ((size_t)(ci0 + r) * taps + t) * Cout + co0 + 8 * c) = make_uint4(u[0], u[1], u[2], u[3]);
        }
    }
}

extern "C" int pai_pack_weights_multi(int n, const float* const* w_master, const int32_t* cout, const int32_t* taps,
                                      const int32_t* cin, void* const* w_fwd, void* const* w_dgrad, void* stream) {
    PAI_CHECK(n > 0 && w_master && cout && taps && cin && w_fwd && w_dgrad, "pai_pack_weights_multi: null pointer");
    for (int i0 = 0; i0 < n; i0 += PACK_MULTI_MAX) {
        PackMulti pm;
        memset(&pm, 0, sizeof(pm));
        pm.n = n - i0 < PACK_MULTI_MAX ? n - i0 : PACK_MULTI_MAX;
        int64_t blocks = 0;
        for (int i = 0; i < pm.n; ++i) {
            const int k = i0 + i;
            PAI_CHECK(w_master[k] && (w_fwd[k] || w_dgrad[k]), "pai_pack_weights_multi: item %d: null pointer", k);
            PAI_CHECK(cin[k] > 0 && cout[k] > 0 && taps[k] > 0 && (cin[k] % 64) == 0 && (cout[k] % 64) == 0,
                      "pai_pack_weights_multi: item %d: Cin=%d, Cout=%d must be multiples of 64 (bf16 packs)", k, cin[k], cout[k]);
            pm.it[i] = PackItem{w_master[k], (bf16_t*)w_fwd[k], (bf16_t*)w_dgrad[k], cout[k], taps[k], cin[k], (int)blocks};
            blocks += (int64_t)(cin[k] / 64) * (cout[k] / 64) * taps[k];
            PAI_CHECK(blocks < (1ll << 31), "pai_pack_weights_multi: too many tiles");
        }
        PAI_LAUNCH(pack64_multi_k, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, pm);
        PAI_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int pai_pack_weights(int dtype, const float* w_master, int Cout, int taps, int Cin,
                                void* w_fwd, void* w_dgrad, void* stream) {
    PAI_CHECK(w_master && (w_fwd || w_dgrad), "pai_pack_weights: null pointer");
    if (dtype == PAI_BF16 && (Cin % 64) == 0 && (Cout % 64) == 0 && Cout / 64 <= 65535 && taps <= 65535) {
        PAI_LAUNCH(pack64_k, dim3(Cin / 64, Cout / 64, taps), dim3(256), 0, (hipStream_t)stream, w_master, Cout,
                           taps, Cin, (bf16_t*)w_fwd, (bf16_t*)w_dgrad);
        PAI_LAUNCH_CHECK();
        return 0;
    }
    dim3 grid(cdiv(Cin, 32), cdiv(Cout, 32), taps);
    if (dtype == PAI_F32)
        PAI_LAUNCH(pack_k<float>, grid, dim3(256), 0, (hipStream_t)stream, w_master, Cout, taps, Cin,
                           (float*)w_fwd, (float*)w_dgrad);
    else
        PAI_LAUNCH(pack_k<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, w_master, Cout, taps, Cin,
                           (bf16_t*)w_fwd, (bf16_t*)w_dgrad);
    PAI_LAUNCH_CHECK();
    return 0;
}

__global__ void reduce_rows_k(const float* partial, int rows, int C, float* out, int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0;
    for (int r = 0; r < rows; ++r) s += (double)partial[(size_t)r * C + c];
    out[c] = accumulate ? out[c] + (float)s : (float)s;
}

extern "C" int pai_reduce_rows(const float* partial, int rows, int C, float* out, int accumulate,
                               void* stream) {
    PAI_CHECK(partial && out, "pai_reduce_rows: null pointer");
    PAI_LAUNCH(reduce_rows_k, dim3(cdiv(C, 64)), dim3(64), 0, (hipStream_t)stream, partial, rows, C,
                       out, accumulate);
    PAI_LAUNCH_CHECK();
    return 0;
}

// out[c] += sum_rows x[row][c]; block = slab of rows, thread = (channel, row lane)
template <typename T>
__global__ __launch_bounds__(256) void colsum_k(const T* x, int64_t rows, int C, int64_t rows_per_block,
                                                float* out) {
    __shared__ float red[256];
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    // blockIdx.y strides over the 256-column chunks (the positional-embedding gradient of the TransUNet is 32 rows x 16384
    // columns: as a loop inside ONE row block that was 64 chunks in sequence, 400 us)
    for (int c0 = blockIdx.y * 256; c0 < C; c0 += gridDim.y * 256) {
        const int width = min(C - c0, 256);
        // largest power of two <= 256/width row lanes
        int lanes = 1;
        while (lanes * 2 * width <= 256) lanes *= 2;
        const int c = tid % width, rl = tid / width;
        float s = 0.f;
        if (rl < lanes)
            for (int64_t r = r0 + rl; r < r1; r += lanes) s += Conv<T>::ld(x + r * C + c0 + c);
        red[tid] = (rl < lanes) ? s : 0.f;
        __syncthreads();
        if (tid < width) {
            float t = 0.f;
            for (int l = 0; l < lanes; ++l) t += red[tid + l * width];
            atomicAdd(out + c0 + tid, t);
        }
        __syncthreads();
    }
}

int launch_colsum(int dtype, const void* x, int64_t rows, int C, float* out, hipStream_t s) {
    int64_t blocks = (rows + 255) / 256;
    if (blocks > 512) blocks = 512;
    if (blocks < 1) blocks = 1;
    const int64_t rpb = (rows + blocks - 1) / blocks;
    blocks = (rows + rpb - 1) / rpb;
    int chunks = cdiv(C, 256);
    if (chunks > 64) chunks = 64;
    if (blocks * chunks > 4096) chunks = (int)(4096 / blocks) > 0 ? (int)(4096 / blocks) : 1;
    const dim3 grid((int)blocks, chunks);
    if (dtype == PAI_F32)
        PAI_LAUNCH(colsum_k<float>, grid, dim3(256), 0, s, (const float*)x, rows, C, rpb, out);
    else
        PAI_LAUNCH(colsum_k<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x, rows, C, rpb, out);
    PAI_LAUNCH_CHECK();
    return 0;
}

extern "C" int pai_colsum(int dtype, const void* x, int64_t rows, int C, float* out, void* stream) {
    PAI_CHECK(dtype == PAI_F32 || dtype == PAI_BF16, "pai_colsum: bad dtype %d", dtype);
    PAI_CHECK(x && out && rows > 0 && C > 0, "pai_colsum: bad arguments");
    return launch_colsum(dtype, x, rows, C, out, (hipStream_t)stream);
}

// One element of torch.optim.Adam (no weight decay / amsgrad), shared by every Adam kernel of the library so that they
// agree bit for bit whatever the compiler would contract in each loop: the fused multiply-adds are spelled out and
// contraction is off.  omb = 1 - beta rounded from double, as torch does.
__device__ __forceinline__ void adam1(float& p, float& m, float& v, float g, float lr_over_bc1, float beta1, float beta2,
                                      float omb1, float omb2, float eps, float inv_sqrt_bc2) {
#pragma clang fp contract(off)
    const float mi = __builtin_fmaf(beta1, m, omb1 * g);
    const float vi = __builtin_fmaf(beta2, v, omb2 * g * g);
    m = mi;
    v = vi;
    p = p - lr_over_bc1 * mi / __builtin_fmaf(sqrtf(vi), inv_sqrt_bc2, eps);
}

__global__ __launch_bounds__(256) void adam_k(float* p, const float* g, float* m, float* v, int64_t numel,
                                              float lr_over_bc1, float beta1, float beta2, float omb1, float omb2,
                                              float eps, float inv_sqrt_bc2) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < numel; i += (int64_t)gridDim.x * 256) {
        adam1(p[i], m[i], v[i], g[i], lr_over_bc1, beta1, beta2, omb1, omb2, eps, inv_sqrt_bc2);
    }
}

extern "C" int pai_adam(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t numel,
                        float lr, float beta1, float beta2, float eps, int step_count, void* stream) {
    PAI_CHECK(param && grad && exp_avg && exp_avg_sq && step_count >= 1, "pai_adam: bad arguments");
    float lr_over_bc1, inv_sqrt_bc2;
    pai::adam_coeffs(lr, beta1, beta2, step_count, &lr_over_bc1, &inv_sqrt_bc2);
    int64_t blocks = (numel + 1023) / 1024;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    pai::plan_mark_adam(5, 11, lr, beta1, beta2, step_count);     // a recorded launch re-derives arguments 5 and 11 per replay
    PAI_LAUNCH(adam_k, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                       exp_avg_sq, numel, lr_over_bc1, beta1, beta2, (float)(1.0 - (double)beta1),
                       (float)(1.0 - (double)beta2), eps, inv_sqrt_bc2);
    PAI_LAUNCH_CHECK();
    return 0;
}

// Adam over one range of the arena that holds ONE dense conv weight ([Cout][taps][Cin], both multiples of 64) plus a few
// small neighbours (its bias, BatchNorm gamma / beta): the weight is walked in the 64 x 64 tiles of pack64_k, so the block
// that has just produced 64 x 64 new fp32 weights also writes their bf16 forward pack and (through LDS) the transposed
// input-gradient pack -- the separate pack launch and its read of every master weight disappear (Pix2Pix generator:
// 218 MB read + one launch on the critical path between two training steps).  Blocks behind the tiles take the
// neighbours elementwise.  Same arithmetic, expression by expression, as adam_k.
__global__ __launch_bounds__(256) void adam_pack64_k(float* p, const float* g, float* m, float* v, int64_t numel,
                                                     int64_t w_off, int Cout, int taps, int Cin, bf16_t* wf, bf16_t* wd,
                                                     int ntiles, float lr_over_bc1, float beta1, float beta2, float omb1,
                                                     float omb2, float eps, float inv_sqrt_bc2) {
    __shared__ float tile[64][65];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= ntiles) {
        const int64_t wn = (int64_t)Cout * taps * Cin;
        const int64_t rest = numel - wn;
        for (int64_t k = (int64_t)(blockIdx.x - ntiles) * 256 + tid; k < rest; k += (int64_t)(gridDim.x - ntiles) * 256) {
            const int64_t i = k < w_off ? k : k + wn;
            adam1(p[i], m[i], v[i], g[i], lr_over_bc1, beta1, beta2, omb1, omb2, eps, inv_sqrt_bc2);
        }
        return;
    }
    const int nci = Cin / 64, nco = Cout / 64;
    const int lb = blockIdx.x;
    const int ci0 = (lb % nci) * 64, co0 = ((lb / nci) % nco) * 64, t = lb / (nci * nco);
    const int q = tid & 15, r0 = tid >> 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = r0 + 16 * j;
        const int64_t i = w_off + ((int64_t)(co0 + r) * taps + t) * Cin + ci0 + 4 * q;
        const float4 g4 = *(const float4*)(g + i);
        float4 m4 = *(const float4*)(m + i), v4 = *(const float4*)(v + i), p4 = *(const float4*)(p + i);
        float* pe = &p4.x; float* me = &m4.x; float* ve = &v4.x; const float* ge = &g4.x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            adam1(pe[e], me[e], ve[e], ge[e], lr_over_bc1, beta1, beta2, omb1, omb2, eps, inv_sqrt_bc2);
            tile[r][4 * q + e] = pe[e];
        }
        *(float4*)(m + i) = m4;
        *(float4*)(v + i) = v4;
        *(float4*)(p + i) = p4;
    }
    __syncthreads();
    const int c = tid & 7, s0 = tid >> 3;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = s0 + 32 * j;
        if (wf) {
            unsigned u[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) u[e] = pk2bf(tile[r][8 * c + 2 * e], tile[r][8 * c + 2 * e + 1]);
            *(uint4*)(wf + ((size_t)(co0 + r) * taps + t) * Cin + ci0 + 8 * c) = make_uint4(u[0], u[1], u[2], u[3]);
        }
        if (wd) {
            unsigned u[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) u[e] = pk2bf(tile[8 * c + 2 * e][r], tile[8 * c + 2 * e + 1][r]);
            *(uint4*)(wd + ((size_t)(ci0 + r) * taps + t) * Cout + co0 + 8 * c) = make_uint4(u[0], u[1], u[2], u[3]);
        }
    }
}

extern "C" int pai_adam_pack(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t numel,
                             int64_t w_off, int Cout, int taps, int Cin, void* w_fwd, void* w_dgrad, float lr,
                             float beta1, float beta2, float eps, int step_count, void* stream) {
    PAI_CHECK(param && grad && exp_avg && exp_avg_sq && step_count >= 1 && (w_fwd || w_dgrad), "pai_adam_pack: bad arguments");
    PAI_CHECK(Cin > 0 && Cout > 0 && taps > 0 && (Cin % 64) == 0 && (Cout % 64) == 0,
              "pai_adam_pack: Cin=%d, Cout=%d must be multiples of 64 (bf16 packs)", Cin, Cout);
    const int64_t wn = (int64_t)Cout * taps * Cin;
    PAI_CHECK(w_off >= 0 && (w_off % 4) == 0 && w_off + wn <= numel, "pai_adam_pack: weight [%lld, %lld) outside the range of %lld",
              (long long)w_off, (long long)(w_off + wn), (long long)numel);
    PAI_CHECK((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq | (uintptr_t)w_fwd | (uintptr_t)w_dgrad) & 15) == 0,
              "pai_adam_pack: pointers must be 16-byte aligned");
    const int64_t ntiles = (int64_t)(Cin / 64) * (Cout / 64) * taps;
    const int64_t rest = numel - wn;
    int64_t extra = (rest + 1023) / 1024;
    if (extra > 256) extra = 256;
    PAI_CHECK(ntiles + extra < (1ll << 31), "pai_adam_pack: too many tiles");
    float lr_over_bc1, inv_sqrt_bc2;
    pai::adam_coeffs(lr, beta1, beta2, step_count, &lr_over_bc1, &inv_sqrt_bc2);
    pai::plan_mark_adam(12, 18, lr, beta1, beta2, step_count);
    PAI_LAUNCH(adam_pack64_k, dim3((unsigned)(ntiles + extra)), dim3(256), 0, (hipStream_t)stream, param, grad,
                       exp_avg, exp_avg_sq, numel, w_off, Cout, taps, Cin, (bf16_t*)w_fwd, (bf16_t*)w_dgrad, (int)ntiles,
                       lr_over_bc1, beta1, beta2, (float)(1.0 - (double)beta1), (float)(1.0 - (double)beta2), eps,
                       inv_sqrt_bc2);
    PAI_LAUNCH_CHECK();
    return 0;
}

// The same update with the step count in DEVICE memory, for a step captured into a hipGraph: a host-side count would be
// frozen into the graph's kernel arguments and every replay would apply the bias correction of the captured step.
// adam_coeff_k advances *step and derives the two step-dependent coefficients (in double, like the host path);
// adam_dev_k is adam_k reading them.
__global__ void adam_coeff_k(long long* step, float* coeff, float lr, float beta1, float beta2) {
    const long long t = *step + 1;
    *step = t;
    const double bc1 = 1.0 - pow((double)beta1, (double)t);
    const double bc2 = 1.0 - pow((double)beta2, (double)t);
    coeff[0] = (float)((double)lr / bc1);
    coeff[1] = (float)(1.0 / sqrt(bc2));
}

__global__ __launch_bounds__(256) void adam_dev_k(float* p, const float* g, float* m, float* v, int64_t numel,
                                                  const float* coeff, float beta1, float beta2, float omb1, float omb2,
                                                  float eps) {
    const float lr_over_bc1 = coeff[0], inv_sqrt_bc2 = coeff[1];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < numel; i += (int64_t)gridDim.x * 256) {
        adam1(p[i], m[i], v[i], g[i], lr_over_bc1, beta1, beta2, omb1, omb2, eps, inv_sqrt_bc2);
    }
}

extern "C" int pai_adam_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t numel,
                            float lr, float beta1, float beta2, float eps, int64_t* step_dev, float* coeff2_dev,
                            void* stream) {
    PAI_CHECK(param && grad && exp_avg && exp_avg_sq && step_dev && coeff2_dev, "pai_adam_dev: null pointer");
    hipStream_t s = (hipStream_t)stream;
    PAI_LAUNCH(adam_coeff_k, dim3(1), dim3(1), 0, s, (long long*)step_dev, coeff2_dev, lr, beta1, beta2);
    PAI_LAUNCH_CHECK();
    int64_t blocks = (numel + 1023) / 1024;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    PAI_LAUNCH(adam_dev_k, dim3((int)blocks), dim3(256), 0, s, param, grad, exp_avg, exp_avg_sq, numel,
                       (const float*)coeff2_dev, beta1, beta2, (float)(1.0 - (double)beta1), (float)(1.0 - (double)beta2), eps);
    PAI_LAUNCH_CHECK();
    return 0;
}

// Multi-tensor form for networks whose parameters are separate allocations (the composable residual / Trans U-Nets):
// up to ADAM_CHUNK tensors per launch, their pointers travelling in the kernel-argument block (no table upload);
// blockIdx.y selects the tensor, blockIdx.x strides over it.
#define ADAM_CHUNK 48
struct AdamChunk {
    float* p[ADAM_CHUNK];
    const float* g[ADAM_CHUNK];
    float* m[ADAM_CHUNK];
    float* v[ADAM_CHUNK];
    int64_t n[ADAM_CHUNK];
};

// One tensor of a multi-tensor launch: 16-byte vectors, two per stream in flight per thread, non-temporal moves for tensors of
// >= 16 MB (Adam touches p, m, v once per step: nothing of it is worth a cache line), a scalar tail.  The arithmetic per
// element is adam1, whatever the access shape.  (The scalar grid-stride loop that stood here moved the TransUNet's 28.9 GB
// per step at 5.2 TB/s.)
typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ void adam_tensor(float* p, const float* g, float* m, float* v, int64_t numel, float lr_over_bc1,
                                            float beta1, float beta2, float omb1, float omb2, float eps, float inv_sqrt_bc2) {
    const bool vec = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0;
    const int64_t n4 = vec ? numel / 4 : 0;
    auto ld = [](const f32x4_t* q) { return NT ? __builtin_nontemporal_load(q) : *q; };
    auto st = [](f32x4_t* q, f32x4_t x) { if (NT) __builtin_nontemporal_store(x, q); else *q = x; };
    const int64_t step = (int64_t)gridDim.x * 512;
    for (int64_t i = (int64_t)blockIdx.x * 512 + threadIdx.x; i < n4; i += step) {
        f32x4_t pv[2], gv[2], mv[2], vv[2];
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (i + k * 256 < n4) {
                pv[k] = ld((const f32x4_t*)p + i + k * 256);
                gv[k] = ld((const f32x4_t*)g + i + k * 256);
                mv[k] = ld((const f32x4_t*)m + i + k * 256);
                vv[k] = ld((const f32x4_t*)v + i + k * 256);
            }
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (i + k * 256 < n4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float pe = pv[k][e], me = mv[k][e], ve = vv[k][e];
                    adam1(pe, me, ve, gv[k][e], lr_over_bc1, beta1, beta2, omb1, omb2, eps, inv_sqrt_bc2);
                    pv[k][e] = pe; mv[k][e] = me; vv[k][e] = ve;
                }
                st((f32x4_t*)p + i + k * 256, pv[k]);
                st((f32x4_t*)m + i + k * 256, mv[k]);
                st((f32x4_t*)v + i + k * 256, vv[k]);
            }
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < numel; i += (int64_t)gridDim.x * 256)
        adam1(p[i], m[i], v[i], g[i], lr_over_bc1, beta1, beta2, omb1, omb2, eps, inv_sqrt_bc2);
}
constexpr int64_t ADAM_NT_NUMEL = 4 << 20;

__global__ __launch_bounds__(256) void adam_multi_k(AdamChunk c, float lr_over_bc1, float beta1, float beta2, float omb1,
                                                    float omb2, float eps, float inv_sqrt_bc2) {
    const int t = blockIdx.y;
    if (c.n[t] >= ADAM_NT_NUMEL)
        adam_tensor<true>(c.p[t], c.g[t], c.m[t], c.v[t], c.n[t], lr_over_bc1, beta1, beta2, omb1, omb2, eps, inv_sqrt_bc2);
    else
        adam_tensor<false>(c.p[t], c.g[t], c.m[t], c.v[t], c.n[t], lr_over_bc1, beta1, beta2, omb1, omb2, eps, inv_sqrt_bc2);
}

__global__ __launch_bounds__(256) void adam_multi_dev_k(AdamChunk c, const float* coeff, float beta1, float beta2, float omb1,
                                                        float omb2, float eps) {
    const float lr_over_bc1 = coeff[0], inv_sqrt_bc2 = coeff[1];
    const int t = blockIdx.y;
    if (c.n[t] >= ADAM_NT_NUMEL)
        adam_tensor<true>(c.p[t], c.g[t], c.m[t], c.v[t], c.n[t], lr_over_bc1, beta1, beta2, omb1, omb2, eps, inv_sqrt_bc2);
    else
        adam_tensor<false>(c.p[t], c.g[t], c.m[t], c.v[t], c.n[t], lr_over_bc1, beta1, beta2, omb1, omb2, eps, inv_sqrt_bc2);
}

// pai_adam_multi with the step count in DEVICE memory (a step captured into a hipGraph, see pai_adam_dev): ONE
// adam_coeff_k advances the count and derives the coefficients, every chunk launch reads them.
extern "C" int pai_adam_multi_dev(int count, void* const* params, const void* const* grads, void* const* exp_avgs,
                                  void* const* exp_avg_sqs, const int64_t* numels, float lr, float beta1, float beta2,
                                  float eps, int64_t* step_dev, float* coeff2_dev, void* stream) {
    PAI_CHECK(count >= 0 && (count == 0 || (params && grads && exp_avgs && exp_avg_sqs && numels)) && step_dev && coeff2_dev,
              "pai_adam_multi_dev: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    PAI_LAUNCH(adam_coeff_k, dim3(1), dim3(1), 0, s, (long long*)step_dev, coeff2_dev, lr, beta1, beta2);
    PAI_LAUNCH_CHECK();
    for (int i0 = 0; i0 < count; i0 += ADAM_CHUNK) {
        const int nt = count - i0 < ADAM_CHUNK ? count - i0 : ADAM_CHUNK;
        AdamChunk c;
        memset(&c, 0, sizeof(c));
        int64_t big = 1;
        for (int i = 0; i < nt; ++i) {
            PAI_CHECK(params[i0 + i] && grads[i0 + i] && exp_avgs[i0 + i] && exp_avg_sqs[i0 + i] && numels[i0 + i] >= 0,
                      "pai_adam_multi_dev: null tensor %d", i0 + i);
            c.p[i] = (float*)params[i0 + i];
            c.g[i] = (const float*)grads[i0 + i];
            c.m[i] = (float*)exp_avgs[i0 + i];
            c.v[i] = (float*)exp_avg_sqs[i0 + i];
            c.n[i] = numels[i0 + i];
            if (c.n[i] > big) big = c.n[i];
        }
        int64_t bx = (big + 2047) / 2048;
        if (bx > 2048) bx = 2048;
        PAI_LAUNCH(adam_multi_dev_k, dim3((unsigned)bx, (unsigned)nt), dim3(256), 0, s, c, (const float*)coeff2_dev,
                           beta1, beta2, (float)(1.0 - (double)beta1), (float)(1.0 - (double)beta2), eps);
        PAI_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int pai_adam_multi(int count, void* const* params, const void* const* grads, void* const* exp_avgs,
                              void* const* exp_avg_sqs, const int64_t* numels, float lr, float beta1, float beta2,
                              float eps, int step_count, void* stream) {
    PAI_CHECK(count >= 0 && (count == 0 || (params && grads && exp_avgs && exp_avg_sqs && numels)) && step_count >= 1,
              "pai_adam_multi: bad arguments");
    float lr_over_bc1, inv_sqrt_bc2;
    pai::adam_coeffs(lr, beta1, beta2, step_count, &lr_over_bc1, &inv_sqrt_bc2);
    for (int i0 = 0; i0 < count; i0 += ADAM_CHUNK) {
        const int nt = count - i0 < ADAM_CHUNK ? count - i0 : ADAM_CHUNK;
        AdamChunk c;
        memset(&c, 0, sizeof(c));
        int64_t big = 1;
        for (int i = 0; i < nt; ++i) {
            PAI_CHECK(params[i0 + i] && grads[i0 + i] && exp_avgs[i0 + i] && exp_avg_sqs[i0 + i] && numels[i0 + i] >= 0,
                      "pai_adam_multi: null tensor %d", i0 + i);
            c.p[i] = (float*)params[i0 + i];
            c.g[i] = (const float*)grads[i0 + i];
            c.m[i] = (float*)exp_avgs[i0 + i];
            c.v[i] = (float*)exp_avg_sqs[i0 + i];
            c.n[i] = numels[i0 + i];
            if (c.n[i] > big) big = c.n[i];
        }
        int64_t bx = (big + 2047) / 2048;
        if (bx > 2048) bx = 2048;
        pai::plan_mark_adam(1, 7, lr, beta1, beta2, step_count);
        PAI_LAUNCH(adam_multi_k, dim3((unsigned)bx, (unsigned)nt), dim3(256), 0, (hipStream_t)stream, c,
                           lr_over_bc1, beta1, beta2, (float)(1.0 - (double)beta1), (float)(1.0 - (double)beta2), eps,
                           inv_sqrt_bc2);
        PAI_LAUNCH_CHECK();
    }
    return 0;
}

// ---- multi-tensor zero fill (gradient-arena segments that are accumulated into) -----------------------------------
#define ZERO_CHUNK 96
struct ZeroChunk {
    float* p[ZERO_CHUNK];
    int64_t n[ZERO_CHUNK];
};

__global__ __launch_bounds__(256) void zero_multi_k(ZeroChunk c) {
    float* p = c.p[blockIdx.y];
    const int64_t n = c.n[blockIdx.y];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = 0.f;
}

extern "C" int pai_zero_multi(int count, void* const* ptrs, const int64_t* numels, void* stream) {
    PAI_CHECK(count >= 0 && (count == 0 || (ptrs && numels)), "pai_zero_multi: bad arguments");
    for (int i0 = 0; i0 < count; i0 += ZERO_CHUNK) {
        const int nt = count - i0 < ZERO_CHUNK ? count - i0 : ZERO_CHUNK;
        ZeroChunk c;
        memset(&c, 0, sizeof(c));
        int64_t big = 1;
        for (int i = 0; i < nt; ++i) {
            PAI_CHECK(ptrs[i0 + i] && numels[i0 + i] >= 0, "pai_zero_multi: null tensor %d", i0 + i);
            c.p[i] = (float*)ptrs[i0 + i];
            c.n[i] = numels[i0 + i];
            if (c.n[i] > big) big = c.n[i];
        }
        int64_t bx = (big + 1023) / 1024;
        if (bx > 256) bx = 256;
        PAI_LAUNCH(zero_multi_k, dim3((unsigned)bx, (unsigned)nt), dim3(256), 0, (hipStream_t)stream, c);
        PAI_LAUNCH_CHECK();
    }
    return 0;
}

// ---- multi-tensor EMA update: shadow -= w * (shadow - param) (reference callbacks/ema.py:24-33, torch_ema) -------------
struct LerpChunk {
    float* d[ADAM_CHUNK];
    const float* s[ADAM_CHUNK];
    int64_t n[ADAM_CHUNK];
};

// torch_ema: tmp = shadow - param; tmp *= (1 - decay); shadow -= tmp -- three roundings.  (__fmul_rn / __fsub_rn are plain
// operators on this target: without the pragma hipcc contracts the last two into one fused multiply-add, 1 ulp off on
// ~1 % of the elements.)
__device__ __forceinline__ float ema1(float sh, float p, float w) {
#pragma clang fp contract(off)
    const float t = (sh - p) * w;
    return sh - t;
}

__global__ __launch_bounds__(256) void lerp_multi_k(LerpChunk c, float w) {
    float* d = c.d[blockIdx.y];
    const float* s = c.s[blockIdx.y];
    const int64_t n = c.n[blockIdx.y];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        d[i] = ema1(d[i], s[i], w);
    }
}

extern "C" int pai_lerp_multi(int count, void* const* dsts, const void* const* srcs, const int64_t* numels, float weight,
                              void* stream) {
    PAI_CHECK(count >= 0 && (count == 0 || (dsts && srcs && numels)), "pai_lerp_multi: bad arguments");
    for (int i0 = 0; i0 < count; i0 += ADAM_CHUNK) {
        const int nt = count - i0 < ADAM_CHUNK ? count - i0 : ADAM_CHUNK;
        LerpChunk c;
        memset(&c, 0, sizeof(c));
        int64_t big = 1;
        for (int i = 0; i < nt; ++i) {
            PAI_CHECK(dsts[i0 + i] && srcs[i0 + i] && numels[i0 + i] >= 0, "pai_lerp_multi: null tensor %d", i0 + i);
            c.d[i] = (float*)dsts[i0 + i];
            c.s[i] = (const float*)srcs[i0 + i];
            c.n[i] = numels[i0 + i];
            if (c.n[i] > big) big = c.n[i];
        }
        int64_t bx = (big + 1023) / 1024;
        if (bx > 2048) bx = 2048;
        PAI_LAUNCH(lerp_multi_k, dim3((unsigned)bx, (unsigned)nt), dim3(256), 0, (hipStream_t)stream, c, weight);
        PAI_LAUNCH_CHECK();
    }
    return 0;
}

// ---- x *= factor over an fp32 buffer (the 1 / world_size average behind a SUM all-reduce) ------------------------------
__global__ __launch_bounds__(256) void scale_k(float* p, int64_t n4, int64_t n, float f) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 v = ((float4*)p)[i];
        v.x *= f; v.y *= f; v.z *= f; v.w *= f;
        ((float4*)p)[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) p[n4 * 4 + threadIdx.x] *= f;
}

extern "C" int pai_scale(float* ptr, int64_t numel, float factor, void* stream) {
    PAI_CHECK(ptr && numel >= 0 && (((uintptr_t)ptr) & 15) == 0, "pai_scale: bad arguments (16-byte aligned fp32 buffer expected)");
    if (numel == 0) return 0;
    int64_t blocks = (numel / 4 + 1023) / 1024;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    PAI_LAUNCH(scale_k, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, ptr, numel / 4, numel, factor);
    PAI_LAUNCH_CHECK();
    return 0;
}

// ---- several dtype casts in one launch (the four NHWC copies in front of a batched discriminator pass) --------------
#define CAST_CHUNK 8
struct CastChunk {
    const void* src[CAST_CHUNK];
    void* dst[CAST_CHUNK];
    int64_t nvec[CAST_CHUNK];      // 8-element vectors per tensor
};

template <typename S, typename D>
__global__ __launch_bounds__(256) void cast_multi_k(CastChunk c) {
    const S* src = (const S*)c.src[blockIdx.y];
    D* dst = (D*)c.dst[blockIdx.y];
    const int64_t n = c.nvec[blockIdx.y];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float v[8];
        V8<S>::ld(src + 8 * i, v);
        V8<D>::st(dst + 8 * i, v);
    }
}

extern "C" int pai_cast_multi(int count, int src_dtype, const void* const* srcs, int dst_dtype, void* const* dsts,
                              const int64_t* numels, void* stream) {
    PAI_CHECK(count >= 0 && count <= CAST_CHUNK && (count == 0 || (srcs && dsts && numels)), "pai_cast_multi: 0..%d tensors", CAST_CHUNK);
    PAI_CHECK((src_dtype == PAI_F32 || src_dtype == PAI_BF16) && (dst_dtype == PAI_F32 || dst_dtype == PAI_BF16),
              "pai_cast_multi: bad dtypes %d -> %d", src_dtype, dst_dtype);
    if (count == 0) return 0;
    CastChunk c;
    memset(&c, 0, sizeof(c));
    int64_t big = 1;
    for (int i = 0; i < count; ++i) {
        PAI_CHECK(srcs[i] && dsts[i] && numels[i] >= 0 && (numels[i] % 8) == 0 &&
                      ((((uintptr_t)srcs[i]) | ((uintptr_t)dsts[i])) & 15) == 0,
                  "pai_cast_multi: tensor %d: null, not a multiple of 8 elements, or not 16-byte aligned", i);
        c.src[i] = srcs[i];
        c.dst[i] = dsts[i];
        c.nvec[i] = numels[i] / 8;
        if (c.nvec[i] > big) big = c.nvec[i];
    }
    int64_t bx = (big + 255) / 256;
    if (bx > 1024) bx = 1024;
    const dim3 grid((unsigned)bx, (unsigned)count);
    hipStream_t s = (hipStream_t)stream;
    if (src_dtype == PAI_F32 && dst_dtype == PAI_BF16) PAI_LAUNCH((cast_multi_k<float, bf16_t>), grid, dim3(256), 0, s, c);
    else if (src_dtype == PAI_BF16 && dst_dtype == PAI_F32) PAI_LAUNCH((cast_multi_k<bf16_t, float>), grid, dim3(256), 0, s, c);
    else if (src_dtype == PAI_F32) PAI_LAUNCH((cast_multi_k<float, float>), grid, dim3(256), 0, s, c);
    else PAI_LAUNCH((cast_multi_k<bf16_t, bf16_t>), grid, dim3(256), 0, s, c);
    PAI_LAUNCH_CHECK();
    return 0;
}


// ---- torch Conv2d filter <-> the library's dense tap-major layout ---------------------------------------------------------
// nn.Conv2d keeps its filter as [Cout][Cin / groups][kh][kw] (reference models/res_unet.py:147-151: groups = 32); the
// convolution entry points take [Cout][kh * kw][Cin] with the groups as diagonal blocks.  One launch each way, on the
// caller's stream, instead of permute / zeros / index_put chains of the tensor library.
__global__ __launch_bounds__(256) void filter_to_dense_k(const float* __restrict__ w, int Cout, int cig, int taps, int groups,
                                                         float* __restrict__ dense) {
    const int Cin = cig * groups, cog = Cout / groups;
    const int64_t total = (int64_t)Cout * taps * Cin;
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {
        const int ci = (int)(o % Cin);
        const int t = (int)((o / Cin) % taps);
        const int co = (int)(o / ((int64_t)Cin * taps));
        const int gi = ci / cig;
        dense[o] = gi == co / cog ? w[((int64_t)co * cig + (ci - gi * cig)) * taps + t] : 0.f;
    }
}

__global__ __launch_bounds__(256) void filter_grad_from_dense_k(const float* __restrict__ dense, int Cout, int cig, int taps,
                                                                int groups, float* __restrict__ dw) {
    const int Cin = cig * groups, cog = Cout / groups;
    const int64_t total = (int64_t)Cout * cig * taps;
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {
        const int t = (int)(o % taps);
        const int cg = (int)((o / taps) % cig);
        const int co = (int)(o / ((int64_t)taps * cig));
        dw[o] = dense[((int64_t)co * taps + t) * Cin + (co / cog) * cig + cg];
    }
}

static int filter_args_ok(const void* a, const void* b, int Cout, int cig, int taps, int groups) {
    return a && b && Cout > 0 && cig > 0 && taps > 0 && groups > 0 && Cout % groups == 0 &&
           (int64_t)Cout * taps * cig * groups < ((int64_t)1 << 31);
}

extern "C" int pai_filter_to_dense(const float* w_oihw, int Cout, int Cin_per_group, int taps, int groups, float* dense,
                                   void* stream) {
    PAI_CHECK(filter_args_ok(w_oihw, dense, Cout, Cin_per_group, taps, groups),
              "pai_filter_to_dense: bad arguments (Cout %d, Cin/groups %d, taps %d, groups %d)", Cout, Cin_per_group, taps, groups);
    const int64_t total = (int64_t)Cout * taps * Cin_per_group * groups;
    int64_t bx = (total + 255) / 256;
    if (bx > 4096) bx = 4096;
    PAI_LAUNCH(filter_to_dense_k, dim3((unsigned)bx), dim3(256), 0, (hipStream_t)stream, w_oihw, Cout, Cin_per_group, taps, groups,
               dense);
    PAI_LAUNCH_CHECK();
    return 0;
}

extern "C" int pai_filter_grad_from_dense(const float* dense_dw, int Cout, int Cin_per_group, int taps, int groups,
                                          float* dw_oihw, void* stream) {
    PAI_CHECK(filter_args_ok(dense_dw, dw_oihw, Cout, Cin_per_group, taps, groups),
              "pai_filter_grad_from_dense: bad arguments (Cout %d, Cin/groups %d, taps %d, groups %d)", Cout, Cin_per_group, taps,
              groups);
    const int64_t total = (int64_t)Cout * taps * Cin_per_group;
    int64_t bx = (total + 255) / 256;
    if (bx > 4096) bx = 4096;
    PAI_LAUNCH(filter_grad_from_dense_k, dim3((unsigned)bx), dim3(256), 0, (hipStream_t)stream, dense_dw, Cout, Cin_per_group,
               taps, groups, dw_oihw);
    PAI_LAUNCH_CHECK();
    return 0;
}


// ---- [A][B][C][D] -> [A][C][B][D] ------------------------------------------------------------------------------------------
// The patch rearrangement in front of and behind the ViT bottleneck (reference models/trans_unet.py:139-141,175-179:
// "n c (h p1) (w p2) -> n (h w) (p1 p2 c)" and back) on NHWC storage is exactly this exchange of the two middle axes
// of [n * grid][p1][grid][p2 * c]; D is moved in 16-byte pieces.
__global__ __launch_bounds__(256) void swap_mid_k(const uint4* __restrict__ src, int64_t rows, int B, int Cc, int Dv,
                                                  uint4* __restrict__ dst) {
    const int64_t total = rows * Dv;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int dv = (int)(i % Dv);
        const int64_t r = i / Dv;               // destination row (a, c, b)
        const int b = (int)(r % B);
        const int c = (int)((r / B) % Cc);
        const int64_t a = r / ((int64_t)B * Cc);
        dst[i] = src[((a * B + b) * Cc + c) * Dv + dv];
    }
}

extern "C" int pai_swap_mid(int elem_bytes, const void* src, int64_t A, int B, int Cc, int64_t D, void* dst, void* stream) {
    PAI_CHECK(src && dst && src != dst && A > 0 && B > 0 && Cc > 0 && D > 0 && (elem_bytes == 2 || elem_bytes == 4),
              "pai_swap_mid: bad arguments");
    PAI_CHECK((D * elem_bytes) % 16 == 0 && ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0 && D * elem_bytes / 16 < (1 << 30),
              "pai_swap_mid: the inner extent (%lld elements of %d bytes) must be a multiple of 16 bytes, 16-byte aligned",
              (long long)D, elem_bytes);
    const int Dv = (int)(D * elem_bytes / 16);
    const int64_t rows = A * B * Cc;
    int64_t bx = (rows * Dv + 255) / 256;
    if (bx > 16384) bx = 16384;
    PAI_LAUNCH(swap_mid_k, dim3((unsigned)bx), dim3(256), 0, (hipStream_t)stream, (const uint4*)src, rows, B, Cc, Dv, (uint4*)dst);
    PAI_LAUNCH_CHECK();
    return 0;
}

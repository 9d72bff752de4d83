// SSIM (Gaussian 11x11, sigma 1.5) + squared-error sums in one pass over the image pair, and the
// gradient of the SSIM / PSNR losses.  Replaces torchmetrics.functional
// structural_similarity_index_measure / peak_signal_noise_ratio / mean_squared_error as called
// from models/utils.py:38-47 (data_range = 1.0) and report.py:78-96,207-212, including the
// denormalisation of models/utils.py:11 in front of them.
//
// The window is applied separably (11 + 11 taps instead of 121) from an LDS tile with a 5-px halo;
// reflect padding supplies the halo at image borders exactly like torchmetrics does, the per-image
// value is the mean over the map cropped by 5 px.
#include <math.h>

#include "common.h"

constexpr int KS = 11, PADW = 5;
constexpr int TS = 32;            // output tile
constexpr int TI = TS + 2 * PADW; // 42 input rows/cols
constexpr float SSIM_C1 = 0.01f * 0.01f, SSIM_C2 = 0.03f * 0.03f;

struct Gauss { float g[KS]; };

static Gauss make_gauss() {
    Gauss k;
    float sum = 0.f;
    for (int i = 0; i < KS; ++i) {
        float d = (float)(i - PADW);
        float v = d / 1.5f;
        k.g[i] = expf(-(v * v) / 2.f);
        sum += k.g[i];
    }
    for (int i = 0; i < KS; ++i) k.g[i] /= sum;
    return k;
}

__device__ __forceinline__ int reflect_idx(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return min(max(i, 0), n - 1);
}

__device__ __forceinline__ float denorm_val(float v, int denorm) {
    return denorm ? fminf(fmaxf(v * 0.5f + 0.5f, 0.f), 1.f) : v;
}

__device__ __forceinline__ void block_add2(double a, double b, double* da, double* db, double* dc) {
    __shared__ double ws[2][4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_xor(a, o, 64);
        b += __shfl_xor(b, o, 64);
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) { ws[0][wid] = a; ws[1][wid] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double sa = ws[0][0] + ws[0][1] + ws[0][2] + ws[0][3];
        const double sb = ws[1][0] + ws[1][1] + ws[1][2] + ws[1][3];
        if (da) atomicAdd(da, sa);
        if (dc) atomicAdd(dc, sa);
        if (db) atomicAdd(db, sb);
    }
}

// MODE 0: forward metrics.  MODE 1: write the three gradient maps (backward pass 1).
template <int MODE>
__global__ __launch_bounds__(256) void ssim_k(const float* pred, const float* target, int H, int W, int denorm,
                                              Gauss gk, double inv_crop, double* out2, double* per_image,
                                              float* full_map, float* dmaps, int NC, float wscale, int xtiles) {
    __shared__ float P[TI][TI + 1], T[TI][TI + 1];
    __shared__ float Hb[5][TI][TS];
    const int tid = threadIdx.x;
    const int nc = blockIdx.z;
    const int y0 = blockIdx.y * TS;
    const float* p = pred + (size_t)nc * H * W;
    const float* t = target + (size_t)nc * H * W;
    // One workgroup walks `xtiles` 32-pixel tiles of its tile row (MODE 0: the whole row, so that the three fp64
    // atomics per workgroup -- all workgroups add to the same two or three words -- happen 512 instead of 4096
    // times per 64-image batch: same-address atomics serialise at the memory side and were most of the 154 us)
    double acc = 0.0, sse = 0.0;
    for (int xt = blockIdx.x * xtiles; xt < (blockIdx.x + 1) * xtiles && xt * TS < W; ++xt) {
    const int x0 = xt * TS;
    if (xt != blockIdx.x * xtiles) __syncthreads();      // the previous tile's readers are done with P, T, Hb
    {
        // all 2 x 7 loads of a thread are issued before the first one is consumed: written as "load, store to LDS,
        // next" the loop was seven dependent HBM round trips per tile
        constexpr int NL = (TI * TI + 255) / 256;
        float pv[NL], tv[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int idx = tid + 256 * i;
            const int r = idx / TI, c = idx - r * TI;
            const int gy = reflect_idx(y0 - PADW + (idx < TI * TI ? r : 0), H), gx = reflect_idx(x0 - PADW + c, W);
            pv[i] = p[(size_t)gy * W + gx];
            tv[i] = t[(size_t)gy * W + gx];
        }
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int idx = tid + 256 * i;
            const int r = idx / TI, c = idx - r * TI;
            if (idx < TI * TI) {
                P[r][c] = denorm_val(pv[i], denorm);
                T[r][c] = denorm_val(tv[i], denorm);
            }
        }
    }
    __syncthreads();
    // horizontal pass, four neighbouring outputs per thread from one 14-value window: a third of the LDS reads and
    // of the products a*a, b*b, a*b of the one-output-per-thread form (the kernel is instruction-bound, not HBM-bound)
    for (int idx = tid; idx < TI * (TS / 4); idx += 256) {
        const int r = idx / (TS / 4), c = (idx - r * (TS / 4)) * 4;
        float wa[KS + 3], wb[KS + 3];
#pragma unroll
        for (int k = 0; k < KS + 3; ++k) { wa[k] = P[r][c + k]; wb[k] = T[r][c + k]; }
        float sp[4] = {0.f, 0.f, 0.f, 0.f}, st[4] = {0.f, 0.f, 0.f, 0.f}, spp[4] = {0.f, 0.f, 0.f, 0.f},
              stt[4] = {0.f, 0.f, 0.f, 0.f}, spt[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < KS + 3; ++k) {
            const float a = wa[k], b = wb[k], aa = a * a, bb = b * b, ab = a * b;
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                if (k - o >= 0 && k - o < KS) {          // same tap order per output as the scalar form: bit-identical sums
                    const float g = gk.g[k - o];
                    sp[o] = fmaf(g, a, sp[o]);
                    st[o] = fmaf(g, b, st[o]);
                    spp[o] = fmaf(g, aa, spp[o]);
                    stt[o] = fmaf(g, bb, stt[o]);
                    spt[o] = fmaf(g, ab, spt[o]);
                }
            }
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            Hb[0][r][c + o] = sp[o]; Hb[1][r][c + o] = st[o]; Hb[2][r][c + o] = spp[o]; Hb[3][r][c + o] = stt[o];
            Hb[4][r][c + o] = spt[o];
        }
    }
    __syncthreads();
    // vertical pass: a thread owns column ox and the four rows 4 (tid >> 5) .. +3, one 14-row window per map
    float vm[5][4];
    {
        const int oyb = (tid >> 5) * 4, ox = tid & 31;
#pragma unroll
        for (int m = 0; m < 5; ++m) {
            float w[KS + 3];
#pragma unroll
            for (int k = 0; k < KS + 3; ++k) w[k] = Hb[m][oyb + k][ox];
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                float acc_ = 0.f;
#pragma unroll
                for (int k = 0; k < KS; ++k) acc_ = fmaf(gk.g[k], w[o + k], acc_);
                vm[m][o] = acc_;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int oy = (tid >> 5) * 4 + j, ox = tid & 31;
        const int y = y0 + oy, x = x0 + ox;
        if (y >= H || x >= W) continue;
        const float mp = vm[0][j], mt = vm[1][j], epp = vm[2][j], ett = vm[3][j], ept = vm[4][j];
        const float spp = epp - mp * mp, stt = ett - mt * mt, spt = ept - mp * mt;
        const float A1 = 2.f * mp * mt + SSIM_C1, A2 = 2.f * spt + SSIM_C2;
        const float B1 = mp * mp + mt * mt + SSIM_C1, B2 = spp + stt + SSIM_C2;
        const float S = (A1 * A2) / (B1 * B2);
        const bool in_crop = y >= PADW && y < H - PADW && x >= PADW && x < W - PADW;
        if (MODE == 0) {
            if (full_map) full_map[((size_t)nc * H + y) * W + x] = S;
            if (in_crop) acc += (double)S;
            const float d = P[oy + PADW][ox + PADW] - T[oy + PADW][ox + PADW];
            sse += (double)d * d;
        } else {
            // d(mean SSIM)/d{mu_p, E[pp], E[pt]} at this window, weighted by the crop mean
            float dmu = 0.f, dpp = 0.f, dpt = 0.f;
            if (in_crop) {
                const float inv = 1.f / (B1 * B2);
                dmu = (2.f * mt * (A2 - A1)) * inv - S * (2.f * mp / B1 - 2.f * mp / B2);
                dpp = -S / B2;
                dpt = 2.f * A1 * inv;
                dmu *= wscale; dpp *= wscale; dpt *= wscale;
            }
            const size_t plane = (size_t)H * W, o = ((size_t)nc * H + y) * W + x;
            dmaps[o] = dmu;
            dmaps[(size_t)NC * plane + o] = dpp;
            dmaps[2 * (size_t)NC * plane + o] = dpt;
        }
    }
    }   // x tiles
    if (MODE == 0)
        block_add2(acc * inv_crop, sse, out2, out2 ? out2 + 1 : nullptr, per_image ? per_image + nc : nullptr);
}

extern "C" int pai_ssim_sse(const float* pred, const float* target, int NC, int H, int W, int denorm,
                            double* out2, double* per_image, float* full_map, void* stream) {
    PAI_CHECK(pred && target && (out2 || per_image || full_map), "pai_ssim_sse: null pointer");
    PAI_CHECK(H > 2 * PADW && W > 2 * PADW, "pai_ssim_sse: image %dx%d smaller than the 11x11 window", H, W);
    static const Gauss gk = make_gauss();
    const double inv_crop = 1.0 / ((double)(H - 2 * PADW) * (double)(W - 2 * PADW));
    // whole tile rows per workgroup once there are enough of them to fill the chip
    const int xt_all = cdiv(W, TS);
    int xtiles = pai_tunable("ssim_rowtiles", 8);
    if (xtiles < 1 || (int64_t)cdiv(H, TS) * NC < 512) xtiles = 1;
    if (xtiles > xt_all) xtiles = xt_all;
    dim3 grid(cdiv(xt_all, xtiles), cdiv(H, TS), NC);
    PAI_LAUNCH(ssim_k<0>, grid, dim3(256), 0, (hipStream_t)stream, pred, target, H, W, denorm, gk,
                       inv_crop, out2, per_image, full_map, (float*)nullptr, NC, 0.f, xtiles);
    PAI_LAUNCH_CHECK();
    return 0;
}

extern "C" int64_t pai_ssim_bwd_workspace_floats(int NC, int H, int W) { return (int64_t)3 * NC * H * W; }

// backward pass 2: grad_i = dp/dpred * [ -(G*Dmu)_i - 2 p_i (G*Dpp)_i - t_i (G*Dpt)_i  (SSIM part)
//                                         + w_psnr * (20/ln10) (p_i - t_i) / SSE ]      (PSNR part)
// G* = zero-padded separable Gaussian correlation (the maps vanish outside the crop).
__global__ __launch_bounds__(256) void ssim_bwd2_k(const float* pred, const float* target, int H, int W,
                                                   int denorm, Gauss gk, const float* dmaps, int NC,
                                                   float w_psnr, const double* sse, float* grad) {
    __shared__ float D[3][TI][TI + 1];
    __shared__ float Hb[3][TI][TS];
    const int tid = threadIdx.x, nc = blockIdx.z;
    const int x0 = blockIdx.x * TS, y0 = blockIdx.y * TS;
    const size_t plane = (size_t)H * W;
    for (int idx = tid; idx < TI * TI; idx += 256) {
        const int r = idx / TI, c = idx - r * TI;
        const int gy = y0 - PADW + r, gx = x0 - PADW + c;
        const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
        const size_t o = ((size_t)nc * H + (in ? gy : 0)) * W + (in ? gx : 0);
#pragma unroll
        for (int q = 0; q < 3; ++q) D[q][r][c] = in ? dmaps[q * (size_t)NC * plane + o] : 0.f;
    }
    __syncthreads();
    for (int idx = tid; idx < TI * TS; idx += 256) {
        const int r = idx / TS, c = idx - r * TS;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const float g = gk.g[k];
            s0 = fmaf(g, D[0][r][c + k], s0);
            s1 = fmaf(g, D[1][r][c + k], s1);
            s2 = fmaf(g, D[2][r][c + k], s2);
        }
        Hb[0][r][c] = s0; Hb[1][r][c] = s1; Hb[2][r][c] = s2;
    }
    __syncthreads();
    const float psnr_k = w_psnr != 0.f ? (float)(w_psnr * (20.0 / log(10.0)) / *sse) : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int oy = (tid >> 5) + 8 * j, ox = tid & 31;
        const int y = y0 + oy, x = x0 + ox;
        if (y >= H || x >= W) continue;
        float f0 = 0.f, f1 = 0.f, f2 = 0.f;
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const float g = gk.g[k];
            f0 = fmaf(g, Hb[0][oy + k][ox], f0);
            f1 = fmaf(g, Hb[1][oy + k][ox], f1);
            f2 = fmaf(g, Hb[2][oy + k][ox], f2);
        }
        const size_t o = ((size_t)nc * H + y) * W + x;
        const float raw = pred[o];
        const float pv = denorm_val(raw, denorm), tv = denorm_val(target[o], denorm);
        float dd = 1.f;
        if (denorm) {
            const float u = raw * 0.5f + 0.5f;
            dd = (u > 0.f && u < 1.f) ? 0.5f : 0.f;   // clamp passes gradient strictly inside (torch.clamp)
            if (u == 0.f || u == 1.f) dd = 0.5f;      // torch: gradient 1 at the boundary values
        }
        const float gs = -(f0 + 2.f * pv * f1 + tv * f2);
        grad[o] = dd * (gs + psnr_k * (pv - tv));
    }
}

extern "C" int pai_ssim_psnr_bwd(const float* pred, const float* target, int NC, int H, int W, int denorm,
                                 float w_ssim, float w_psnr, const double* sse, float* grad,
                                 float* workspace, void* stream) {
    PAI_CHECK(pred && target && grad && workspace, "pai_ssim_psnr_bwd: null pointer");
    PAI_CHECK(w_psnr == 0.f || sse, "pai_ssim_psnr_bwd: PSNR term needs sse");
    PAI_CHECK(H > 2 * PADW && W > 2 * PADW, "pai_ssim_psnr_bwd: image too small");
    static const Gauss gk = make_gauss();
    const double crop = (double)(H - 2 * PADW) * (double)(W - 2 * PADW);
    // loss = -(w_ssim * mean over NC planes of the crop mean + w_psnr * PSNR)
    const float wscale = (float)((double)w_ssim / (crop * (double)NC));
    dim3 grid(cdiv(W, TS), cdiv(H, TS), NC);
    hipStream_t s = (hipStream_t)stream;
    PAI_LAUNCH(ssim_k<1>, grid, dim3(256), 0, s, pred, target, H, W, denorm, gk, 0.0,
                       (double*)nullptr, (double*)nullptr, (float*)nullptr, workspace, NC, wscale, 1);
    PAI_LAUNCH_CHECK();
    PAI_LAUNCH(ssim_bwd2_k, grid, dim3(256), 0, s, pred, target, H, W, denorm, gk, workspace, NC,
                       w_psnr, sse, grad);
    PAI_LAUNCH_CHECK();
    return 0;
}

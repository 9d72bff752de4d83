// Column-owner finish kernels of the split-K launches whose OUTPUT is small (<= 4096 rows: the U-Net bottleneck,
// encoders[4-6] / decoders[0-2] of BASELINE configs[1] and the input gradients that feed them).
//
// The general path finishes a split-K layer in a chain of dependent launches of 5-13 us each:
//   forward:   gg_fwd_mfma_k<split> -> splitk_finish_k (slab sum, bias, bf16 z, per-16-row BatchNorm partials)
//              -> bn_finalize_wide_k (mean / rstd / running statistics) -> bn_apply_k (normalise + activation)
//   backward:  gg_fwd_mfma_k<split> -> splitk_finish_k (slab sum, fused producer backward, per-tile partials)
//              -> bn_bwd_finalize_k (sum du, sum du xhat, dgamma, dbeta) -> bn_bwd_apply_k (dz)
// The statistics need the whole batch, which is what forces the launch boundaries -- unless one workgroup OWNS a
// channel group over ALL rows.  With <= 4096 rows that is cheap: a workgroup of 256 threads takes 8 channels, every
// thread sums the K-split slabs of its <= 16 rows (two 16-B loads per row and slab), the 8 channel statistics meet in
// the workgroup (fp64, deterministic), and the same threads normalise / back-propagate the values they still hold in
// registers.  One launch instead of three, no intermediate tensor (z is written once, du never), 14 launches fewer on
// the critical path of a training step.  STATUS (round 3): correct and tested, NOT faster -- see finish_fused_ok below.
//
// Serves nn.BatchNorm2d(train) + activation behind the bottleneck convolutions (reference models/pix2pix.py:63-70,99-106)
// and the matching half of aten::native_batch_norm_backward.
#include "gg_tile.h"

constexpr int FB_ROWS = 16;          // rows per thread
constexpr int FB_MAX_ROWS = 256 * FB_ROWS;

__device__ __forceinline__ void fb_block_sums(const float* s1, const float* s2, double (*red)[4][8], int tid, double* t1, double* t2) {
    // 8 per-thread fp32 partials -> fp64 totals of the workgroup (wave: shuffles, then the 4 waves through LDS)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        double a = (double)s1[k], b = (double)s2[k];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
        if ((tid & 63) == 0) { red[0][tid >> 6][k] = a; red[1][tid >> 6][k] = b; }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        t1[k] = red[0][0][k] + red[0][1][k] + red[0][2][k] + red[0][3][k];
        t2[k] = red[1][0][k] + red[1][1][k] + red[1][2][k] + red[1][3][k];
    }
}

// Sum of the K-split slabs for the rows of this workgroup's 8 channels (slab layout [split][channel group][row][8],
// written by gg_fwd_mfma_k when FwdArgs.skip_finish is set: 32 B per thread, contiguous across the threads).  Thread (rl = tid % RL, sl = tid / RL): RL =
// min(256, rows rounded up to a power of two) row lanes, SL = 256 / RL split lanes.  A thread owns rows rl + RL i and the
// splits sl, sl + SL, ..; per split ALL its rows are requested before any is added (up to 32 independent 16-B loads in
// flight per thread: with the split loop innermost a thread waited for memory once per four splits and row, and the
// launch was a chain of ~30 memory round trips).  With SL > 1 (fewer than 256 rows: the 2 x 2 and 1 x 1 layers, whose
// cost model picks 32-64 splits) the split lanes meet through LDS, in lane order (deterministic); the result is then
// valid in the threads sl == 0 only.
template <int NR>
__device__ __forceinline__ void fb_slab_sum(const float* ws, int ksplit, int rows, int Cout, int c0, int tid, int RL,
                                            float (*v)[8], float* xch /* LDS [256][8] or NULL when SL == 1 */) {
    const int SL = 256 / RL, rl = tid & (RL - 1), sl = tid / RL;
    const size_t slab = (size_t)rows * Cout;
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
        for (int k = 0; k < 8; ++k) v[i][k] = 0.f;
    constexpr int NB = NR < 8 ? NR : 8;               // rows requested together (16 independent 16-B loads per thread)
    for (int s = sl; s < ksplit; s += SL) {
#pragma unroll
        for (int i0 = 0; i0 < NR; i0 += NB) {
            float4 t0[NB], t1[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int r = rl + RL * (i0 + i);
                if (r < rows) {
                    const float* src = ws + (size_t)s * slab + ((size_t)(c0 >> 3) * rows + r) * 8;   // [split][group][row][8]
                    t0[i] = *(const float4*)src;
                    t1[i] = *(const float4*)(src + 4);
                } else {
                    t0[i] = t1[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                float* o = v[i0 + i];
                o[0] += t0[i].x; o[1] += t0[i].y; o[2] += t0[i].z; o[3] += t0[i].w;
                o[4] += t1[i].x; o[5] += t1[i].y; o[6] += t1[i].z; o[7] += t1[i].w;
            }
        }
    }
    if (SL > 1) {      // NR == 1 here: one row per row lane
#pragma unroll
        for (int k = 0; k < 8; ++k) xch[tid * 8 + k] = v[0][k];
        __syncthreads();
        if (sl == 0) {
            for (int q = 1; q < SL; ++q)
#pragma unroll
                for (int k = 0; k < 8; ++k) v[0][k] += xch[(q * RL + rl) * 8 + k];
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void splitk_finish_bn_k(GG g, FinishBnArgs f, int RL) {
    __shared__ double red[2][4][8];
    __shared__ float coef[2][8];
    const int tid = threadIdx.x;
    const int c0 = blockIdx.x * 8;
    const int rows = g.nphase * g.M;
    const size_t slab = (size_t)rows * g.Cout;
    __shared__ float xch[256 * 8];
    float v[FB_ROWS][8], s1[8], s2[8], bv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { s1[k] = s2[k] = 0.f; bv[k] = f.bias ? f.bias[c0 + k] : 0.f; }
    const bool owner = tid < RL;                       // threads that hold finished rows (all of them when RL == 256)
    if (RL == 256) fb_slab_sum<FB_ROWS>(f.ws, f.ksplit, rows, g.Cout, c0, tid, 256, v, nullptr);
    else fb_slab_sum<1>(f.ws, f.ksplit, rows, g.Cout, c0, tid, RL, v, xch);
#pragma unroll
    for (int i = 0; i < FB_ROWS; ++i) {
        const int r = tid + 256 * i;                   // RL < 256: only i == 0 holds a row (r = tid < RL)
        if ((RL == 256 || i == 0) && owner && r < rows) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                v[i][k] += bv[k];
                s1[k] += v[i][k];
                s2[k] = fmaf(v[i][k], v[i][k], s2[k]);
            }
        }
    }
    double t1[8], t2[8];
    fb_block_sums(s1, s2, red, tid, t1, t2);
    if (blockIdx.x == 0 && tid == 0 && f.nbt) *f.nbt += f.n_updates;
    if (tid < 8) {
        // same arithmetic as bn_finalize_wide_k (bn.hip)
        const int c = c0 + tid;
        const double count = (double)rows;
        const double mean = t1[tid] / count;
        double var = t2[tid] / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)f.eps));
        const float gm = f.gamma ? f.gamma[c] : 1.f, bt = f.beta ? f.beta[c] : 0.f;
        const float sc = gm * rstd, sh = bt - (float)mean * sc;
        f.mean[c] = (float)mean;
        f.rstd[c] = rstd;
        f.scale[c] = sc;
        f.shift[c] = sh;
        coef[0][tid] = sc;
        coef[1][tid] = sh;
        if (f.running_mean && f.running_var) {
            const float unbiased = (float)(count > 1.0 ? var * count / (count - 1.0) : var);
            float rm = f.running_mean[c], rv = f.running_var[c];
            for (int u = 0; u < f.n_updates; ++u) {
                rm = (1.f - f.momentum) * rm + f.momentum * (float)mean;
                rv = (1.f - f.momentum) * rv + f.momentum * unbiased;
            }
            f.running_mean[c] = rm;
            f.running_var[c] = rv;
        }
    }
    __syncthreads();
    float sc[8], sh[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { sc[k] = coef[0][k]; sh[k] = coef[1][k]; }
#pragma unroll
    for (int i = 0; i < FB_ROWS; ++i) {
        const int r = tid + 256 * i;
        if (r >= rows || !owner || (RL < 256 && i > 0)) continue;
        const int ph = r / g.M, m = r - ph * g.M;
        int n, gy, gx;
        decode_row(g, m, n, gy, gx);
        const size_t off = ((size_t)(n * g.OH + gy * g.OS + g.poy[ph]) * g.OW + gx * g.OS + g.pox[ph]) * g.Cout + c0;
        unsigned zp[4], ap[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            zp[k] = pk2bf(v[i][2 * k], v[i][2 * k + 1]);
            // the activated tensor is formed from the bf16-rounded z, as pai_bn_apply forms it from the stored z
            const float z0 = __uint_as_float(zp[k] << 16), z1 = __uint_as_float(zp[k] & 0xffff0000u);
            ap[k] = pk2bf(act_apply(fmaf(z0, sc[2 * k], sh[2 * k]), f.act), act_apply(fmaf(z1, sc[2 * k + 1], sh[2 * k + 1]), f.act));
        }
        if (f.z) *(uint4*)(f.z + off) = make_uint4(zp[0], zp[1], zp[2], zp[3]);
        *(uint4*)(f.a + off) = make_uint4(ap[0], ap[1], ap[2], ap[3]);
    }
}

// blockIdx.x < D1 / 8: 8 channels of the first destination, through the producer's activation and BatchNorm backward;
// the rest: 8 channels of the second destination (the skip path), plain bf16 store.
__global__ __launch_bounds__(256) void splitk_finish_bnbwd_k(GG g, FwdArgs a, FinishBwdArgs f, int RL) {
    __shared__ double red[2][4][8];
    __shared__ float coef[2][8];
    const int tid = threadIdx.x;
    const int rows = g.nphase * g.M;
    const bool first = (int)blockIdx.x * 8 < g.D1;
    const int c0 = blockIdx.x * 8;                       // column of the slab
    const int cd = first ? c0 : c0 - g.D1;               // column of the destination tensor
    const int dstride = first ? g.D1 : g.D2;
    const bf16_t* bzp = (const bf16_t*)a.bz;
    const bf16_t* bap = (const bf16_t*)a.badd;
    BwdParams BP;
    if (first) bwd_load_params(a, cd, BP);
    __shared__ float xch[256 * 8];
    uint4 du[FB_ROWS], zq[FB_ROWS];
    float s1[8], s2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s1[k] = s2[k] = 0.f;
    float v[FB_ROWS][8];
    const bool owner = tid < RL;
    if (RL == 256) fb_slab_sum<FB_ROWS>(f.ws, f.ksplit, rows, g.Cout, c0, tid, 256, v, nullptr);
    else fb_slab_sum<1>(f.ws, f.ksplit, rows, g.Cout, c0, tid, RL, v, xch);
#pragma unroll
    for (int i = 0; i < FB_ROWS; ++i) {
        const int r = tid + 256 * i;
        du[i] = zq[i] = make_uint4(0, 0, 0, 0);
        if (r >= rows || !owner || (RL < 256 && i > 0)) continue;
        const uint4 o = make_uint4(pk2bf(v[i][0], v[i][1]), pk2bf(v[i][2], v[i][3]), pk2bf(v[i][4], v[i][5]), pk2bf(v[i][6], v[i][7]));
        const int ph = r / g.M, m = r - ph * g.M;
        int n, gy, gx;
        decode_row(g, m, n, gy, gx);
        const size_t off = ((size_t)(n * g.OH + gy * g.OS + g.poy[ph]) * g.OW + gx * g.OS + g.pox[ph]) * dstride + cd;
        if (!first) {
            *(uint4*)((bf16_t*)a.y2 + off) = o;
            continue;
        }
        zq[i] = *(const uint4*)(bzp + off);
        const uint4 aq = bap ? *(const uint4*)(bap + off) : make_uint4(0, 0, 0, 0);
        // same du as splitk_finish_k writes (bf16, from the bf16-rounded gradient), sums from the value as rounded
        du[i] = bwd_chunk(o, zq[i], aq, bap != nullptr, a.bscale != nullptr, true, a.bact1, a.bact2, BP, s1, s2);
    }
    if (!first) return;
    double t1[8], t2[8];
    fb_block_sums(s1, s2, red, tid, t1, t2);
    if (tid < 8) {
        const int c = cd + tid;
        const float S1 = (float)t1[tid];
        const float S2 = (float)((double)a.brstd[c] * (t2[tid] - (double)a.bmean[c] * t1[tid]));   // sum du * xhat from sum du * z
        coef[0][tid] = S1;
        coef[1][tid] = S2;
        if (f.sums) { f.sums[c] = S1; f.sums[g.D1 + c] = S2; }
        if (f.dbeta) f.dbeta[c] += S1;
        if (f.dgamma) f.dgamma[c] += S2;
    }
    __syncthreads();
    // dz = gamma * rstd * (du - S1 / M - xhat * S2 / M)      (bn_bwd_apply_k, same expression)
    const float inv_m = (float)(1.0 / (double)rows);
    float mu[8], rs[8], gm[8], sb[8], sg[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        mu[k] = a.bmean[cd + k];
        rs[k] = a.brstd[cd + k];
        gm[k] = f.gamma ? f.gamma[cd + k] : 1.f;
        sb[k] = coef[0][k];
        sg[k] = coef[1][k];
    }
#pragma unroll
    for (int i = 0; i < FB_ROWS; ++i) {
        const int r = tid + 256 * i;
        if (r >= rows || !owner || (RL < 256 && i > 0)) continue;
        const int ph = r / g.M, m = r - ph * g.M;
        int n, gy, gx;
        decode_row(g, m, n, gy, gx);
        const size_t off = ((size_t)(n * g.OH + gy * g.OS + g.poy[ph]) * g.OW + gx * g.OS + g.pox[ph]) * g.D1 + cd;
        const unsigned dw[4] = {du[i].x, du[i].y, du[i].z, du[i].w}, zw[4] = {zq[i].x, zq[i].y, zq[i].z, zq[i].w};
        unsigned o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float d[2], z[2];
            d[0] = __uint_as_float(dw[k] << 16); d[1] = __uint_as_float(dw[k] & 0xffff0000u);
            z[0] = __uint_as_float(zw[k] << 16); z[1] = __uint_as_float(zw[k] & 0xffff0000u);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int c = 2 * k + e;
                const float xh = (z[e] - mu[c]) * rs[c];
                d[e] = gm[c] * rs[c] * (d[e] - sb[c] * inv_m - xh * sg[c] * inv_m);
            }
            o[k] = pk2bf(d[0], d[1]);
        }
        *(uint4*)(f.dz + off) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// ---- host ---------------------------------------------------------------------------------------------------------
// the split-K launch of this problem is followed by a column-owner finish: few rows, 8-channel groups
bool finish_fused_ok(const GG& g, int d1_cols) {
    // OFF by default (tunable finish_fused = 1 turns it on): measured in the BASELINE configs[1] step on one box, three
    // interleaved pairs: 6.64 / 6.65 / 6.65 ms per step with the fused finish against 6.55 / 6.56 / 6.52 with the three
    // launches (first version, row-major slabs read 32 B per row: 7.02 against 6.70).  A column owner is 64 workgroups
    // (Cout / 8), one per CU at 290 registers: the 8-33 MB of slabs of a bottleneck layer then stream through a quarter
    // of the chip, while splitk_finish_k spreads them over 1024 workgroups; the two launch boundaries saved (~3 us) do
    // not pay for that.  Kept: bit-exact (tests/test_gpu_finish.py), and the two entry points are the right place for
    // a better fusion.
    if (!pai_tunable("finish_fused", 0)) return false;
    if (fwd_mfma_ksplit_effective(g) <= 1) return false;
    return (int64_t)g.nphase * g.M <= FB_MAX_ROWS && (g.Cout % 8) == 0 && (d1_cols % 8) == 0;
}

static int fb_row_lanes(const GG& g) {      // row lanes of a workgroup: the rows rounded up to a power of two, at most 256
    const int rows = g.nphase * g.M;
    int rl = 1;
    while (rl < rows && rl < 256) rl <<= 1;
    return rl;
}

int launch_finish_bn(const GG& g, const FinishBnArgs& f, hipStream_t s) {
    PAI_LAUNCH(splitk_finish_bn_k, dim3(g.Cout / 8), dim3(256), 0, s, g, f, fb_row_lanes(g));
    PAI_LAUNCH_CHECK();
    return 0;
}

int launch_finish_bnbwd(const GG& g, const FwdArgs& a, const FinishBwdArgs& f, hipStream_t s) {
    PAI_LAUNCH(splitk_finish_bnbwd_k, dim3(g.Cout / 8), dim3(256), 0, s, g, a, f, fb_row_lanes(g));
    PAI_LAUNCH_CHECK();
    return 0;
}

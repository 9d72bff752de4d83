// Small-channel convolutions on the matrix cores: 1x1 and 3x3 "same" Conv2d (forward and input gradient) where one
// side has 16 or 32 channels -- the bottlenecks of the TransUNet / ResNet-50 encoder blocks
// (reference models/trans_unet.py:203-227: in/4 channels; models/res_unet.py:86-95).  The tile kernels of gg_mfma.hip
// need 64-channel K steps and 64-wide output tiles; these layers used to fall to the vector-ALU kernel
// (12 TFLOP/s) although they are HBM-bound (a 16 -> 16 3x3 at 256 x 256 x 32 images is 10 GFLOP against 134 MB).
//
// No LDS.  A wave keeps the whole filter slice of its output-channel group in registers as the MFMA A operand
// (K = taps x Cin in steps of 32, zero-padded) and streams groups of 16 output pixels: per K step a lane loads the 8
// consecutive input channels (16 B) of ONE tap of its pixel -- Cin is a multiple of 8, so a 16-B piece never
// straddles taps -- then NT MFMAs; after the K loop a lane holds 4 consecutive output channels of its pixel per
// 16-channel tile.  Bias, BatchNorm partial statistics (one row per workgroup) and the activation in the store.
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf8_t;
typedef __attribute__((ext_vector_type(4))) float f4_t;
typedef __attribute__((ext_vector_type(8))) short s8_t;

__device__ __forceinline__ bf8_t small_relu8(bf8_t f) {
    s8_t x = __builtin_bit_cast(s8_t, f);
    const s8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
    return __builtin_bit_cast(bf8_t, __builtin_elementwise_max(x, z));
}

static int small_ksteps(const GG& g) { return cdiv(g.ntaps * g.Cin, 32); }

// output-channel tiles (of 16) per workgroup: as many as the register budget of the filter slice allows
static int small_ntb(const GG& g) {
    const int ks = small_ksteps(g), nt = g.Cout / 16;
    const int cap = ks <= 2 ? 8 : (ks <= 4 ? 4 : 2);
    int b = 1;
    while (b * 2 <= cap && (nt % (b * 2)) == 0) b *= 2;
    return b;
}

bool small_ok(int dtype, const GG& g, const FwdArgs& a) {
    if (dtype != PAI_BF16) return false;
    if (g.nphase != 1 || g.S != 1 || g.OS != 1 || (g.ntaps != 1 && g.ntaps != 9)) return false;
    if (g.C2 != 0 || g.D2 != 0 || a.yf32 || a.skip_d1 || a.bz) return false;
    if ((g.C1 % 8) || (g.Cout % 16)) return false;
    // the tile kernels' territory (measured: pointwise 64 / 128-channel layers at 4 M pixels run 3-38 % slower here)
    if ((g.C1 % 64) == 0 && (g.Cout % 64) == 0) return false;
    const int ks = small_ksteps(g);
    if (!(ks == 1 || ks == 2 || ks == 4 || ks == 5 || ks == 8 || ks == 9)) return false;
    if (a.yact && a.eact != PAI_ACT_NONE && a.eact != PAI_ACT_LRELU && a.eact != PAI_ACT_RELU) return false;
    return true;
}

static int small_blocks(const GG& g) {
    const int groups = cdiv(g.M, 16);
    int b = cdiv(groups, 4);
    return b > 2048 ? 2048 : b;
}

int small_rows(const GG& g) { return small_blocks(g); }

template <int KS, int NT>
__global__ __launch_bounds__(256) void small_fwd_k(GG g, FwdArgs a, int groups_per_wave) {
    __shared__ float sred[4][2][NT * 16];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int co0 = blockIdx.y * (NT * 16);
    const bf16_t* x = (const bf16_t*)a.x1;
    const bf16_t* w = (const bf16_t*)a.w;
    const int K = g.ntaps * g.Cin;

    // A operand: Wp[co][wt[t]][ci] for row co = co0 + 16 nt + fr, k = 32 s + 8 fq .. + 7 (zero beyond K)
    bf8_t af[NT][KS];
    int ddy[KS], ddx[KS], cofs[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int kk = 32 * s + 8 * fq;
        const bool kv = kk < K;
        const int t = kv ? kk / g.Cin : 0;
        const int ci = kv ? kk - t * g.Cin : 0;
        ddy[s] = g.dy[0][t];
        ddx[s] = kv ? g.dx[0][t] : -(1 << 20);          // padding columns of K: always outside the image
        cofs[s] = ci;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            uint4 z = make_uint4(0, 0, 0, 0);
            if (kv) z = *(const uint4*)(w + ((size_t)(co0 + 16 * nt + fr) * g.wtaps + g.wt[0][t]) * g.Cin + ci);
            af[nt][s] = __builtin_bit_cast(bf8_t, z);
        }
    }
    float bias[NT][4], csum[NT][4], csq[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            bias[nt][r] = a.bias ? a.bias[co0 + 16 * nt + 4 * fq + r] : 0.f;
            csum[nt][r] = csq[nt][r] = 0.f;
        }
    const float eslope = act_slope(a.yact ? a.eact : PAI_ACT_NONE);      // branch-free activation (common.h)
    bf16_t* yraw = (bf16_t*)a.y1;
    bf16_t* yact = (bf16_t*)a.yact;

    const int g0 = (blockIdx.x * 4 + wid) * groups_per_wave;
    for (int gi = 0; gi < groups_per_wave; ++gi) {
        const int m = (g0 + gi) * 16 + fr;
        if ((g0 + gi) * 16 >= g.M) break;                 // uniform per wave
        const bool mv = m < g.M;
        int n, gy, gx;
        if (g.lw >= 0) {
            gx = m & (g.OWg - 1);
            gy = (m >> g.lw) & (g.OHg - 1);
            n = m >> (g.lw + g.lh);
        } else {
            gx = m % g.OWg;
            const int rr = m / g.OWg;
            gy = rr % g.OHg;
            n = rr / g.OHg;
        }
        bf8_t bfr[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int iy = gy + ddy[s], ix = gx + ddx[s];
            const bool inb = mv && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
            // clamped address + select: the KS loads are issued back to back
            const size_t off = inb ? ((size_t)(n * g.H + iy) * g.W + ix) * g.Cin + cofs[s] : 0;
            uint4 v = *(const uint4*)(x + off);
            if (!inb) v = make_uint4(0, 0, 0, 0);
            bfr[s] = __builtin_bit_cast(bf8_t, v);
            if (g.relu1) bfr[s] = small_relu8(bfr[s]);
        }
        f4_t acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = (f4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[nt][s], bfr[s], acc[nt], 0, 0, 0);
        // D[i = 4 fq + r][j = fr]: channel co0 + 16 nt + 4 fq + r of pixel m
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = acc[nt][r] + bias[nt][r];
                if (mv) {
                    csum[nt][r] += v[r];
                    csq[nt][r] = fmaf(v[r], v[r], csq[nt][r]);
                }
            }
            if (!mv) continue;
            const size_t o = (size_t)m * g.Cout + co0 + 16 * nt + 4 * fq;
            if (yraw) *(uint2*)(yraw + o) = make_uint2(pk2bf(v[0], v[1]), pk2bf(v[2], v[3]));
            if (yact) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = act_fwd(v[r], eslope);
                *(uint2*)(yact + o) = make_uint2(pk2bf(v[0], v[1]), pk2bf(v[2], v[3]));
            }
        }
    }
    if (!a.stats) return;
    // per-workgroup partial statistics row: sum over the 16 pixel lanes of every channel, then over the 4 waves
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float s1 = csum[nt][r], s2 = csq[nt][r];
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                s1 += __shfl_xor(s1, o, 64);
                s2 += __shfl_xor(s2, o, 64);
            }
            if (fr == 0) {
                sred[wid][0][16 * nt + 4 * fq + r] = s1;
                sred[wid][1][16 * nt + 4 * fq + r] = s2;
            }
        }
    __syncthreads();
    if (tid < NT * 16) {
        float* dst = a.stats + ((size_t)blockIdx.x * 2) * g.Cout + co0 + tid;
        dst[0] = sred[0][0][tid] + sred[1][0][tid] + sred[2][0][tid] + sred[3][0][tid];
        dst[g.Cout] = sred[0][1][tid] + sred[1][1][tid] + sred[2][1][tid] + sred[3][1][tid];
    }
}

int launch_small(const GG& g, const FwdArgs& a, hipStream_t s) {
    const int ks = small_ksteps(g), ntb = small_ntb(g);
    const int bx = small_blocks(g);
    const int gpw = cdiv(cdiv(g.M, 16), bx * 4);
    const dim3 grid(bx, g.Cout / (16 * ntb));
#define SMALL_LAUNCH(KS_, NT_) PAI_LAUNCH((small_fwd_k<KS_, NT_>), grid, dim3(256), 0, s, g, a, gpw)
#define SMALL_NT(KS_, MAXNT)                                         \
    do {                                                             \
        if (ntb == 1) SMALL_LAUNCH(KS_, 1);                          \
        else if (ntb == 2) SMALL_LAUNCH(KS_, 2);                     \
        else if (ntb == 4 && MAXNT >= 4) SMALL_LAUNCH(KS_, (MAXNT >= 4 ? 4 : 1)); \
        else if (ntb == 8 && MAXNT >= 8) SMALL_LAUNCH(KS_, (MAXNT >= 8 ? 8 : 1)); \
        else { pai_set_error("small conv: no instantiation for ks=%d ntb=%d", ks, ntb); return 1; } \
    } while (0)
    switch (ks) {
        case 1: SMALL_NT(1, 8); break;
        case 2: SMALL_NT(2, 8); break;
        case 4: SMALL_NT(4, 4); break;
        case 5: SMALL_NT(5, 2); break;
        case 8: SMALL_NT(8, 2); break;
        case 9: SMALL_NT(9, 2); break;
        default: pai_set_error("small conv: unsupported K steps %d", ks); return 1;
    }
#undef SMALL_NT
#undef SMALL_LAUNCH
    PAI_LAUNCH_CHECK();
    return 0;
}

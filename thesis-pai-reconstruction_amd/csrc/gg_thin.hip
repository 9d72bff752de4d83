// "Thin" layers of the hot path on the matrix cores: convolutions with <= 2 channels on one
// side and 64/128 on the other (bf16 storage):
//   encoders[0]  Conv2d(1, 64)        models/pix2pix.py:141-147        (thin -> wide)
//   D block 0    Conv2d(1|1, 64)      models/wrapper.py:229,237        (thin -> wide)
//   decoders[7]  ConvTranspose2d(64|64, 1) + tanh  models/pix2pix.py:185-196   (wide -> thin)
// and their gradients.  These layers are HBM-bound (they touch a 64/128-channel tensor once), but a
// straightforward per-pixel dot product is issue-bound on the vector ALUs (1.5 ms for the head at
// batch 64).  Here each one is a skinny GEMM on v_mfma_f32_16x16x32_bf16 with the wide tensor
// streamed exactly once:
//   TF  thin -> wide forward      D[co][pix]  = sum_k W[co][k] * patch[pix][k]        (K = 16*T <= 32)
//   TD  wide -> thin, two steps   Y[pix][t,tap] = sum_c Wp[t,tap][c] * X[pix][c]      (N = 16*T)
//                                 out[2a+ph][2b+pw][t] = bias + sum of 4 Y entries     (col2im)
//   TW  thin weight gradient      D[k][wc]    = sum_pix patch[pix][k] * wide[pix][wc] (K = pixels)
// patch[pix][k=(tap,t)] = thin_t[n][S*gy + dy[tap]][S*gx + dx[tap]] (zero outside the image).
#include <stdlib.h>

#include "common.h"

#ifndef THIN_ABL
#define THIN_ABL 0     // compile-time timing ablations of thin_fwd_k (results WRONG): 1 no gathers, 2 no stores
#endif
typedef __attribute__((ext_vector_type(8))) __bf16 bf8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf4_t;
typedef __attribute__((ext_vector_type(4))) float f4_t;
typedef __attribute__((ext_vector_type(8))) short s8_t;
typedef __attribute__((ext_vector_type(8))) unsigned short us8_t;

__device__ uint4 g_zero_line_thin[16];  // 256 B of zeros (device symbols are per translation unit)


__device__ __forceinline__ bf8_t relu8(bf8_t f) {
    s8_t x = __builtin_bit_cast(s8_t, f);
    const s8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
    return __builtin_bit_cast(bf8_t, __builtin_elementwise_max(x, z));
}

// Eight elements thin[e0 + ix0 + S * j], j = 0..7 (S = 1 or 2) of one row of a 1-channel bf16 image, as two / one
// 16-B buffer loads + byte permutes instead of eight 2-byte gathers with their own index arithmetic (the gathers were
// 60 % of D block 0's weight gradient).  e0 = element index of the row start; the row is `valid` or reads as zero;
// elements left of 0 or right of TW - 1 read as zero.  The wide loads start at the even element below ix0.
template <int S>
__device__ __forceinline__ us8_t thin_gather8(__amdgpu_buffer_rsrc_t rs, int e0, int ix0, int TW, bool valid) {
    typedef __attribute__((ext_vector_type(4))) unsigned u4_t;
    const int base = ix0 & ~1, o = ix0 & 1;
    // the very first row of the tensor, left edge: the wide load would start one dword BEFORE the tensor and come back
    // all zero (the range check is per instruction, not per dword) -- start at 0 and shift by a dword instead
    const bool neg = valid && e0 + base < 0;
    const unsigned vo = valid ? (neg ? 0u : (unsigned)(e0 + base) * 2u) : 0x80000000u;
    unsigned out[4];
    if (S == 2) {
        u4_t a = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)vo, 0, 0);
        u4_t b = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)vo + 16, 0, 0);
        if (__builtin_amdgcn_ballot_w64(neg) != 0) {
            if (neg) { b = (u4_t){a[3], b[0], b[1], b[2]}; a = (u4_t){0u, a[0], a[1], a[2]}; }
        }
        const unsigned sel = o ? 0x07060302u : 0x05040100u;      // half o of the low dword | half o of the high dword
        out[0] = __builtin_amdgcn_perm(a[1], a[0], sel);
        out[1] = __builtin_amdgcn_perm(a[3], a[2], sel);
        out[2] = __builtin_amdgcn_perm(b[1], b[0], sel);
        out[3] = __builtin_amdgcn_perm(b[3], b[2], sel);
    } else {
        u4_t a = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)vo, 0, 0);
        unsigned a4 = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)vo + 16, 0, 0);
        if (__builtin_amdgcn_ballot_w64(neg) != 0) {
            if (neg) { a4 = a[3]; a = (u4_t){0u, a[0], a[1], a[2]}; }
        }
        const unsigned sh = (unsigned)o * 16u;
        out[0] = __builtin_amdgcn_alignbit(a[1], a[0], sh);
        out[1] = __builtin_amdgcn_alignbit(a[2], a[1], sh);
        out[2] = __builtin_amdgcn_alignbit(a[3], a[2], sh);
        out[3] = __builtin_amdgcn_alignbit(a4, a[3], sh);
    }
    // edges (the first / last pixels of an image row): elements outside [0, TW) read as zero; wave-uniform branch
    const bool edge = valid && (ix0 < 0 || ix0 + S * 7 >= TW);
    if (__builtin_amdgcn_ballot_w64(edge) != 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool lo = (unsigned)(ix0 + S * (2 * i)) < (unsigned)TW, hi = (unsigned)(ix0 + S * (2 * i + 1)) < (unsigned)TW;
            out[i] &= (lo ? 0xffffu : 0u) | (hi ? 0xffff0000u : 0u);
        }
    }
    us8_t r;
#pragma unroll
    for (int i = 0; i < 4; ++i) { r[2 * i] = (unsigned short)(out[i] & 0xffffu); r[2 * i + 1] = (unsigned short)(out[i] >> 16); }
    return r;
}

// ReLU on two packed bf16: as signed 16-bit integers every negative float is negative
__device__ __forceinline__ unsigned relu2u(unsigned v) {
    typedef __attribute__((ext_vector_type(2))) short s2_t;
    const s2_t z = {0, 0};
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s2_t, v), z));
}

// Four CONTIGUOUS elements thin[e0 + ix0 .. + 3] of one row (the four kx taps of one kernel row at one pixel) as one
// 12-B buffer load + two funnel shifts; same conventions as thin_gather8.  Returns two packed pairs.
__device__ __forceinline__ uint2 thin_gather4(__amdgpu_buffer_rsrc_t rs, int e0, int ix0, int TW, bool valid) {
    typedef __attribute__((ext_vector_type(3))) unsigned u3_t;
    const int base = ix0 & ~1, o = ix0 & 1;
    const bool neg = valid && e0 + base < 0;
    const unsigned vo = valid ? (neg ? 0u : (unsigned)(e0 + base) * 2u) : 0x80000000u;
    u3_t a = __builtin_amdgcn_raw_buffer_load_b96(rs, (int)vo, 0, 0);
    if (__builtin_amdgcn_ballot_w64(neg) != 0) {
        if (neg) a = (u3_t){0u, a[0], a[1]};
    }
    const unsigned sh = (unsigned)o * 16u;
    uint2 out = make_uint2(__builtin_amdgcn_alignbit(a[1], a[0], sh), __builtin_amdgcn_alignbit(a[2], a[1], sh));
    const bool edge = valid && (ix0 < 0 || ix0 + 3 >= TW);
    if (__builtin_amdgcn_ballot_w64(edge) != 0) {
        out.x &= ((unsigned)ix0 < (unsigned)TW ? 0xffffu : 0u) | ((unsigned)(ix0 + 1) < (unsigned)TW ? 0xffff0000u : 0u);
        out.y &= ((unsigned)(ix0 + 2) < (unsigned)TW ? 0xffffu : 0u) | ((unsigned)(ix0 + 3) < (unsigned)TW ? 0xffff0000u : 0u);
    }
    return out;
}

__device__ __forceinline__ void decode_row2(const GG& g, int m, int& n, int& gy, int& gx) {
    if (g.lw >= 0) {
        gx = m & (g.OWg - 1);
        gy = (m >> g.lw) & (g.OHg - 1);
        n = m >> (g.lw + g.lh);
    } else {
        gx = m % g.OWg;
        int r = m / g.OWg;
        gy = r % g.OHg;
        n = r / g.OHg;
    }
}

// ------------------------------------------------------------------------------------------------
// TF: thin -> wide forward (conv form: one phase, 16 taps).  No LDS: the filter tile lives in
// registers, the patch fragment is gathered from the (L1/L2-resident) 1-channel images.
// ------------------------------------------------------------------------------------------------
bool thin_fwd_ok(int dtype, const GG& g, const FwdArgs& a) {
    // 16 taps: the k4 convolutions of the Pix2Pix nets; 9 taps: the 3x3 in_conv of the residual / Trans U-Nets
    // (models/res_unet.py:265, models/trans_unet.py:66) and the input gradient of their 64 -> 1 out convolution
    const bool taps_ok = g.ntaps == 16 || (g.ntaps == 9 && g.S == 1 && g.C2 == 0);
    return dtype == PAI_BF16 && g.nphase == 1 && taps_ok && g.OS == 1 && g.C1 == 1 && g.C2 <= 1 &&
           (g.Cout % 64) == 0 && g.Cout <= 128 && !a.stats && !a.yf32 && !a.skip_d1 &&
           (g.D2 == 0 || (g.D1 % 16) == 0);
}

template <int T>  // thin channels (1 or 2, one per source tensor)
__global__ __launch_bounds__(256) void thin_fwd_k(GG g, FwdArgs a, int fast_ok) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const bf16_t* x1 = (const bf16_t*)a.x1;
    const bf16_t* x2 = (const bf16_t*)a.x2;
    const bf16_t* w = (const bf16_t*)a.w;
    const int mtiles = g.Cout / 16;
    const int KT = g.ntaps * T;      // 16 T, or 9 for the 3x3 layers (zero-padded to the 32 of the MFMA)
    // bit 1 of fast_ok: every store INSTRUCTION covers whole 64-B sectors.  A lane holds 16 channels of one pixel; as
    // one 32-B run (two stores to consecutive addresses) each instruction writes 16 of every 32 bytes -- half of every
    // sector.  With the lane's channels split into the 8 at 8 fq and the 8 at 32 + 8 fq, the four lanes of a pixel
    // write its first 64 B with the first instruction and its second 64 B with the second.
    const bool sect = (fast_ok & 2) != 0;
    fast_ok &= 1;

    // A operand: W[co][k], k = tap*T + t, zero beyond 16*T.  MFMA row i of tile mt computes output
    // channel co(mt, i) = 64*(mt>>2) + 16*(i>>2) + 4*(mt&3) + (i&3): after the four tiles of a group a
    // lane (fq = i>>2 of its D rows) holds 16 CONSECUTIVE channels of its pixel -> two 16-B stores
    // per pixel and group instead of four scattered 8-B ones.
    bf8_t af[8];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
        us8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
        // (sect: the lane's 16 channels are two 8-channel pieces 64 B apart -- see the stores)
        const int co = sect ? 64 * (mt >> 2) + 32 * ((mt & 3) >> 1) + 8 * (fr >> 2) + 4 * (mt & 1) + (fr & 3)
                            : 64 * (mt >> 2) + 16 * (fr >> 2) + 4 * (mt & 3) + (fr & 3);
        if (mt < mtiles) {
            if ((KT & 7) == 0) {
                if (8 * fq < KT) z = *(const us8_t*)(w + (size_t)co * KT + 8 * fq);
            } else {                    // 9-element filter rows: no 16-B alignment, element by element (once per wave)
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (8 * fq + e < KT) z[e] = w[(size_t)co * KT + 8 * fq + e];
            }
        }
        af[mt] = __builtin_bit_cast(bf8_t, z);
    }
    // bias of this lane's 16 channels per group: co = 64*grp + 16*fq + e
    float bias[2][16];
#pragma unroll
    for (int gq = 0; gq < 2; ++gq)
#pragma unroll
        for (int e = 0; e < 16; ++e)
            bias[gq][e] = (a.bias && gq * 4 < mtiles) ? a.bias[64 * gq + (sect ? 32 * (e >> 3) + 8 * fq + (e & 7) : 16 * fq + e)] : 0.f;
    // this lane's 8 patch elements: (tap, t) pairs
    int pdy[8], pdx[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * fq + j;
        const int tap = (k < KT) ? k / T : 0;
        pdy[j] = g.dy[0][tap];
        pdx[j] = (k < KT) ? g.dx[0][tap] : -100000;   // padding columns of K: always out of bounds
    }

    // The patch gathers of group i + 1 are issued before group i is multiplied and stored: in program order
    // (gather -> MFMA -> store) the 2-byte loads and the 16-B stores of a wave never overlapped (THIN_ABL: 19 us of
    // loop, +17 us gathers, +14 us stores on encoders[0]).
    auto gather = [&](int p0) {
        const int m = p0 + fr;
        int n, gy, gx;
        decode_row2(g, m < g.M ? m : 0, n, gy, gx);
        us8_t pv;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int iy = gy * g.S + pdy[j], ix = gx * g.S + pdx[j];
            const bool inb = m < g.M && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
            const size_t off = (size_t)(n * g.H + iy) * g.W + ix;
            const bf16_t* src = (T == 2 && (j & 1)) ? x2 : x1;
            unsigned short v = (THIN_ABL & 1) ? (unsigned short)(off & 0x3fff) : (inb ? src[off] : (unsigned short)0);
            if (inb && ((T == 2 && (j & 1)) ? g.relu2 : g.relu1) && (v & 0x8000)) v = 0;
            pv[j] = v;
        }
        return pv;
    };
    // 4 x 4 kernels (tap = 4 ky + kx): a lane's eight patch elements are the four kx taps of two kernel rows (T = 1) or
    // of one kernel row in both source tensors (T = 2) -- contiguous in the source row: two 12-B loads (thin_gather4)
    // instead of eight 2-byte gathers.  The 3 x 3 layers keep the element-wise path.
    const bool fast = fast_ok && g.ntaps == 16;
    const unsigned thin_bytes = (unsigned)(g.N * g.H * g.W) * 2u;
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(x1), 0, thin_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(T == 2 ? x2 : x1), 0, thin_bytes, 0x00020000);
    const bool lane_k = 8 * fq < KT;             // this lane's k range holds taps at all (T = 1: fq < 2)
    auto gather_fast = [&](int p0) {
        const int m = p0 + fr;
        int n, gy, gx;
        decode_row2(g, m < g.M ? m : 0, n, gy, gx);
        const int ix0 = gx * g.S + pdx[0];
        const int iyA = gy * g.S + pdy[0], iyB = gy * g.S + pdy[4];
        const bool okA = lane_k && m < g.M && (unsigned)iyA < (unsigned)g.H;
        uint2 pa, pb;
        if (T == 1) {
            const bool okB = lane_k && m < g.M && (unsigned)iyB < (unsigned)g.H;
            pa = thin_gather4(rs1, (n * g.H + iyA) * g.W, ix0, g.W, okA);
            pb = thin_gather4(rs1, (n * g.H + iyB) * g.W, ix0, g.W, okB);
            if (g.relu1) {
                pa.x = relu2u(pa.x);
                pa.y = relu2u(pa.y);
                pb.x = relu2u(pb.x);
                pb.y = relu2u(pb.y);
            }
        } else {
            uint2 qa = thin_gather4(rs1, (n * g.H + iyA) * g.W, ix0, g.W, okA);
            uint2 qb = thin_gather4(rs2, (n * g.H + iyA) * g.W, ix0, g.W, okA);
            if (g.relu1) {
                qa.x = relu2u(qa.x);
                qa.y = relu2u(qa.y);
            }
            if (g.relu2) {
                qb.x = relu2u(qb.x);
                qb.y = relu2u(qb.y);
            }
            // k = 2 tap + t: (x1 kx0, x2 kx0, x1 kx1, x2 kx1 | x1 kx2, ...)
            pa = make_uint2(__builtin_amdgcn_perm(qb.x, qa.x, 0x05040100u), __builtin_amdgcn_perm(qb.x, qa.x, 0x07060302u));
            pb = make_uint2(__builtin_amdgcn_perm(qb.y, qa.y, 0x05040100u), __builtin_amdgcn_perm(qb.y, qa.y, 0x07060302u));
        }
        return __builtin_bit_cast(us8_t, make_uint4(pa.x, pa.y, pb.x, pb.y));
    };
    const float act_slope = a.eact == PAI_ACT_LRELU ? 0.2f : 1.f;
    const float act_floor = a.eact == PAI_ACT_RELU ? 0.f : -__builtin_inff();
    const int pstep = gridDim.x * 64;
    int p0 = (blockIdx.x * 4 + wid) * 16;
    us8_t pv = {0, 0, 0, 0, 0, 0, 0, 0};
    if (p0 < g.M) pv = fast ? gather_fast(p0) : gather(p0);
    for (; p0 < g.M; p0 += pstep) {
        const int m = p0 + fr;
        us8_t pnext = {0, 0, 0, 0, 0, 0, 0, 0};
        if (p0 + pstep < g.M) pnext = fast ? gather_fast(p0 + pstep) : gather(p0 + pstep);
        const bf8_t bfrag = __builtin_bit_cast(bf8_t, pv);
        // rows beyond M carry an all-zero patch; the MFMAs run unconditionally (full wave), only the
        // stores are predicated
#pragma unroll
        for (int gq = 0; gq < 2; ++gq) {
            if (gq * 4 >= mtiles) break;
            float v[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f4_t acc = {0.f, 0.f, 0.f, 0.f};
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[gq * 4 + q], bfrag, acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[4 * q + r] = acc[r] + bias[gq][4 * q + r];
            }
            if (m >= g.M) continue;
            if ((THIN_ABL & 2) && v[0] != 12345.f) continue;   // ablation: no stores
            const int co = 64 * gq + (sect ? 8 : 16) * fq;   // first of this lane's 16 channels
            const int hop = sect ? 32 : 8;                   // channel distance of the lane's second 8-channel piece
            const size_t pix = (size_t)m;
            if (a.y1 || a.y2) {
                unsigned pk[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) pk[e] = pk2bf(v[2 * e], v[2 * e + 1]);
                bf16_t* dst = (co < g.D1) ? (bf16_t*)a.y1 + pix * g.D1 + co : (bf16_t*)a.y2 + pix * g.D2 + (co - g.D1);
                *(uint4*)dst = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                *(uint4*)(dst + hop) = make_uint4(pk[4], pk[5], pk[6], pk[7]);
            }
            if (a.yact) {
                unsigned pk[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    // branch-free (the activation is a run-time value: as if / else per element this unrolled loop was
                    // ~50 scalar branches per 16 pixels): LeakyReLU max(v, 0.2 v), ReLU max(v, 0), none max(v, -inf)
                    // (a NaN pre-activation stays a NaN, as in aten: fmaxf(NaN, x) = x would have laundered it into
                    //  -inf / 0 and hidden a diverged run from isfinite checks -- the clamp is a compare + select)
                    float v0 = fmaxf(v[2 * e], act_slope * v[2 * e]), v1 = fmaxf(v[2 * e + 1], act_slope * v[2 * e + 1]);
                    v0 = v0 < act_floor ? act_floor : v0;
                    v1 = v1 < act_floor ? act_floor : v1;
                    pk[e] = pk2bf(v0, v1);
                }
                bf16_t* dst = (bf16_t*)a.yact + pix * g.Cout + co;
                *(uint4*)dst = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                *(uint4*)(dst + hop) = make_uint4(pk[4], pk[5], pk[6], pk[7]);
            }
        }
        pv = pnext;
    }
}

// Round 4: the same computation with the run-time switches of the store path compiled away and the per-value vector work
// cut from ~5 to ~2.5 instructions (the loop was as long as the store stream and the two did not overlap, DESIGN.md
// section 10): the bias rides in as the MFMA's C operand, the activation is a template parameter (LeakyReLU = mul + max,
// ReLU = max, none = nothing; no floor / slope registers), stores go through buffer descriptors with 32-bit offsets that
// advance by a constant per iteration (rows beyond M fall outside the descriptor and are dropped: no predication), and
// the two 8-channel pieces of a lane always form whole 64-B sectors.  4 x 4 kernels with even row length only
// (thin_gather4); everything else stays on thin_fwd_k.
//   RAW: write the un-activated output (y1 | y2);  ACTM: 0 no activated output, 1 LeakyReLU(0.2), 2 ReLU, 3 identity
//   BWD (RAW, D1 = 64): the first pass of the BatchNorm backward of the layer that PRODUCED the y1 tensor rides on the
//   store (pai_conv_dgrad_bn with act1 = none, no second gradient: du IS the value stored): z of the same elements is
//   read beside the store and one partial row [2][64] = (sum du, sum du * xhat) is written per workgroup -- decoders[6]
//   behind the head's input gradient: bn_bwd_reduce_k (88 us in the step: 268 MB read) becomes 134 MB read here.
#ifndef THIN_BWD_WAVES
#define THIN_BWD_WAVES 1
#endif
template <int T, bool RAW, int ACTM, int NG, bool BWD = false>      // NG: 64-channel groups (Cout / 64)
__global__ __launch_bounds__(256, BWD ? THIN_BWD_WAVES : 1) void thin_fwd2_k(GG g, FwdArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const bf16_t* w = (const bf16_t*)a.w;
    constexpr int mtiles = 4 * NG;
    constexpr int KT = 16 * T;
    bf8_t af[4 * NG];
#pragma unroll
    for (int mt = 0; mt < 4 * NG; ++mt) {
        us8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
        const int co = 64 * (mt >> 2) + 32 * ((mt & 3) >> 1) + 8 * (fr >> 2) + 4 * (mt & 1) + (fr & 3);
        if (mt < mtiles && 8 * fq < KT) z = *(const us8_t*)(w + (size_t)co * KT + 8 * fq);
        af[mt] = __builtin_bit_cast(bf8_t, z);
    }
    // bias as the accumulator the MFMA starts from: acc[r] of tile q is channel 64 gq + 32 (q >> 1) + 8 fq + 4 (q & 1) + r
    f4_t bias4[NG][4];
#pragma unroll
    for (int gq = 0; gq < NG; ++gq)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                // (BWD = an input gradient: no bias; the zero accumulator is then an inline constant, not 32 registers)
                bias4[gq][q][r] = (!BWD && a.bias) ? a.bias[64 * gq + 32 * (q >> 1) + 8 * fq + 4 * (q & 1) + r] : 0.f;
    const int pdx0 = g.dx[0][(8 * fq < KT) ? (8 * fq) / T : 0];
    const int pdyA = g.dy[0][(8 * fq < KT) ? (8 * fq) / T : 0];
    const int pdyB = g.dy[0][(8 * fq + 4 < KT) ? (8 * fq + 4) / T : 0];
    const unsigned thin_bytes = (unsigned)(g.N * g.H * g.W) * 2u;
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x1), 0, thin_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(T == 2 ? a.x2 : a.x1), 0, thin_bytes, 0x00020000);
    const bool lane_k = 8 * fq < KT;
    auto gather_fast = [&](int p0) {
        const int m = p0 + fr;
        int n, gy, gx;
        decode_row2(g, m < g.M ? m : 0, n, gy, gx);
        const int ix0 = gx * g.S + pdx0;
        const int iyA = gy * g.S + pdyA, iyB = gy * g.S + pdyB;
        const bool okA = lane_k && m < g.M && (unsigned)iyA < (unsigned)g.H;
        uint2 pa, pb;
        if (T == 1) {
            const bool okB = lane_k && m < g.M && (unsigned)iyB < (unsigned)g.H;
            pa = thin_gather4(rs1, (n * g.H + iyA) * g.W, ix0, g.W, okA);
            pb = thin_gather4(rs1, (n * g.H + iyB) * g.W, ix0, g.W, okB);
            if (g.relu1) { pa.x = relu2u(pa.x); pa.y = relu2u(pa.y); pb.x = relu2u(pb.x); pb.y = relu2u(pb.y); }
        } else {
            uint2 qa = thin_gather4(rs1, (n * g.H + iyA) * g.W, ix0, g.W, okA);
            uint2 qb = thin_gather4(rs2, (n * g.H + iyA) * g.W, ix0, g.W, okA);
            if (g.relu1) { qa.x = relu2u(qa.x); qa.y = relu2u(qa.y); }
            if (g.relu2) { qb.x = relu2u(qb.x); qb.y = relu2u(qb.y); }
            pa = make_uint2(__builtin_amdgcn_perm(qb.x, qa.x, 0x05040100u), __builtin_amdgcn_perm(qb.x, qa.x, 0x07060302u));
            pb = make_uint2(__builtin_amdgcn_perm(qb.y, qa.y, 0x05040100u), __builtin_amdgcn_perm(qb.y, qa.y, 0x07060302u));
        }
        return __builtin_bit_cast(us8_t, make_uint4(pa.x, pa.y, pb.x, pb.y));
    };
    // output descriptors: 32-bit byte offsets, rows >= M are out of range (dropped by the hardware)
    const unsigned raw1_bytes = (unsigned)g.M * (unsigned)g.D1 * 2u, raw2_bytes = (unsigned)g.M * (unsigned)g.D2 * 2u;
    const __amdgpu_buffer_rsrc_t y1rs = __builtin_amdgcn_make_buffer_rsrc(a.y1, 0, RAW ? raw1_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t y2rs = __builtin_amdgcn_make_buffer_rsrc(a.y2 ? a.y2 : a.y1, 0, (RAW && a.y2) ? raw2_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t yars = __builtin_amdgcn_make_buffer_rsrc(a.yact, 0, ACTM ? (unsigned)g.M * (unsigned)g.Cout * 2u : 0u, 0x00020000);
    typedef __attribute__((ext_vector_type(4))) unsigned u4_t;
    const __amdgpu_buffer_rsrc_t zrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.bz ? a.bz : a.x1), 0, (BWD && RAW) ? raw1_bytes : 0u, 0x00020000);
    float bs1[BWD ? 16 : 1], bs2[BWD ? 16 : 1];      // this lane's 16 channels of group 0: sum du, sum du * z
#pragma unroll
    for (int c = 0; c < (BWD ? 16 : 1); ++c) bs1[c] = bs2[c] = 0.f;
    const int pstep = gridDim.x * 64;
    int p0 = (blockIdx.x * 4 + wid) * 16;
    // this lane's byte offsets of its first 8-channel piece (group 0) in each output
    unsigned o_act = (unsigned)(p0 + fr) * (unsigned)g.Cout * 2u + 16u * fq;
    unsigned o_r1 = (unsigned)(p0 + fr) * (unsigned)g.D1 * 2u + 16u * fq;
    unsigned o_r2 = (unsigned)(p0 + fr) * (unsigned)g.D2 * 2u + 16u * fq;
    const unsigned s_act = (unsigned)pstep * (unsigned)g.Cout * 2u, s_r1 = (unsigned)pstep * (unsigned)g.D1 * 2u,
                   s_r2 = (unsigned)pstep * (unsigned)g.D2 * 2u;
    const bool second_raw = RAW && g.D2 > 0;      // group 1 of a 128-channel layer goes to y2 (D1 = D2 = 64), else to y1 + 128 B
    // The patch gathers run THREE iterations ahead (4 registers each): an iteration is one round trip to L2 / HBM
    // (~2-3 us under load), each CU holds 6-7 waves per SIMD and every wave has only ~18 iterations -- one iteration of
    // lookahead left the loop latency-bound (D block 0 at batch 128: 81 us warm against 40 us of stores).
    constexpr int AHEAD = 3;
    us8_t pq[AHEAD];
#pragma unroll
    for (int i = 0; i < AHEAD; ++i) {
        pq[i] = (us8_t){0, 0, 0, 0, 0, 0, 0, 0};
        if ((int64_t)p0 + (int64_t)i * pstep < g.M) pq[i] = gather_fast(p0 + i * pstep);
    }
    // BWD: z of the elements an iteration stores, requested ZA iterations ahead like the gathers (round 6: the request used
    // to sit at the top of the iteration that consumes it -- one exposed HBM round trip per iteration, 185 us against 56 us
    // without the fused sums for decoders[7]'s input gradient).  rows >= M: out of range, read as zero (and the value stored
    // for them is dropped); offsets beyond the tensor likewise.
#ifndef THIN_ZA
#define THIN_ZA 2
#endif
    constexpr int ZA = THIN_ZA;
    u4_t zql[BWD ? ZA : 1], zqh[BWD ? ZA : 1];
    if (BWD) {
#pragma unroll
        for (int i = 0; i < ZA; ++i) {
            zql[i] = __builtin_amdgcn_raw_buffer_load_b128(zrs, (int)(o_r1 + (unsigned)i * s_r1), 0, 0);
            zqh[i] = __builtin_amdgcn_raw_buffer_load_b128(zrs, (int)(o_r1 + (unsigned)i * s_r1) + 64, 0, 0);
        }
    }
    for (; p0 < g.M; p0 += pstep) {
        us8_t pnext = {0, 0, 0, 0, 0, 0, 0, 0};
        if ((int64_t)p0 + (int64_t)AHEAD * pstep < g.M) pnext = gather_fast(p0 + AHEAD * pstep);
        const bf8_t bfrag = __builtin_bit_cast(bf8_t, pq[0]);
        u4_t zlo = {0u, 0u, 0u, 0u}, zhi = {0u, 0u, 0u, 0u};
        if (BWD) {
            zlo = zql[0]; zhi = zqh[0];
#pragma unroll
            for (int i = 0; i + 1 < ZA; ++i) { zql[i] = zql[i + 1]; zqh[i] = zqh[i + 1]; }
            // (an offset past 2^32 cannot occur: out_ok bounds M x Cout x 2 by 2^32 and ZA x s_r1 is added to a valid row)
            const unsigned long long zo = (unsigned long long)o_r1 + (unsigned long long)ZA * s_r1;
            const int zoff = zo < (unsigned long long)raw1_bytes ? (int)(unsigned)zo : (int)0x80000000u;
            zql[ZA - 1] = __builtin_amdgcn_raw_buffer_load_b128(zrs, zoff, 0, 0);
            zqh[ZA - 1] = __builtin_amdgcn_raw_buffer_load_b128(zrs, zoff + 64, 0, 0);
        }
#pragma unroll
        for (int gq = 0; gq < NG; ++gq) {
            f4_t acc[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[gq * 4 + q], bfrag, bias4[gq][q], 0, 0, 0);
            if (RAW) {
                const u4_t lo = {pk2bf(acc[0][0], acc[0][1]), pk2bf(acc[0][2], acc[0][3]), pk2bf(acc[1][0], acc[1][1]), pk2bf(acc[1][2], acc[1][3])};
                const u4_t hi = {pk2bf(acc[2][0], acc[2][1]), pk2bf(acc[2][2], acc[2][3]), pk2bf(acc[3][0], acc[3][1]), pk2bf(acc[3][2], acc[3][3])};
                if (BWD && gq == 0 && p0 + fr < g.M) {
                    // the bf16 values as stored (what a second pass would read back), against z of the same elements
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float g0 = __uint_as_float(lo[e] << 16), g1 = __uint_as_float(lo[e] & 0xffff0000u);
                        const float h0 = __uint_as_float(hi[e] << 16), h1 = __uint_as_float(hi[e] & 0xffff0000u);
                        bs1[2 * e] += g0; bs1[2 * e + 1] += g1; bs1[8 + 2 * e] += h0; bs1[8 + 2 * e + 1] += h1;
                        bs2[2 * e] = fmaf(g0, __uint_as_float(zlo[e] << 16), bs2[2 * e]);
                        bs2[2 * e + 1] = fmaf(g1, __uint_as_float(zlo[e] & 0xffff0000u), bs2[2 * e + 1]);
                        bs2[8 + 2 * e] = fmaf(h0, __uint_as_float(zhi[e] << 16), bs2[8 + 2 * e]);
                        bs2[8 + 2 * e + 1] = fmaf(h1, __uint_as_float(zhi[e] & 0xffff0000u), bs2[8 + 2 * e + 1]);
                    }
                }
                if (gq == 1 && second_raw) {
                    __builtin_amdgcn_raw_buffer_store_b128(lo, y2rs, (int)o_r2, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(hi, y2rs, (int)o_r2 + 64, 0, 0);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b128(lo, y1rs, (int)o_r1 + 128 * gq, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(hi, y1rs, (int)o_r1 + 128 * gq + 64, 0, 0);
                }
            }
            if (ACTM) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (ACTM == 1) acc[q][r] = fmaxf(acc[q][r], 0.2f * acc[q][r]);
                        else if (ACTM == 2) acc[q][r] = acc[q][r] < 0.f ? 0.f : acc[q][r];      // NaN stays NaN (aten's relu)
                    }
                const u4_t lo = {pk2bf(acc[0][0], acc[0][1]), pk2bf(acc[0][2], acc[0][3]), pk2bf(acc[1][0], acc[1][1]), pk2bf(acc[1][2], acc[1][3])};
                const u4_t hi = {pk2bf(acc[2][0], acc[2][1]), pk2bf(acc[2][2], acc[2][3]), pk2bf(acc[3][0], acc[3][1]), pk2bf(acc[3][2], acc[3][3])};
                __builtin_amdgcn_raw_buffer_store_b128(lo, yars, (int)o_act + 128 * gq, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(hi, yars, (int)o_act + 128 * gq + 64, 0, 0);
            }
        }
        o_act += s_act; o_r1 += s_r1; o_r2 += s_r2;
#pragma unroll
        for (int i = 0; i + 1 < AHEAD; ++i) pq[i] = pq[i + 1];
        pq[AHEAD - 1] = pnext;
    }
    if (BWD) {
        // sum over the wave's 16 pixel lanes (fr: one DPP row), then over the four waves through LDS; channel of
        // accumulator c of row fq: 8 fq + c (c < 8), 32 + 8 fq + (c - 8) otherwise
        __shared__ float red[4][2][64];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            float s1 = bs1[c], s2 = bs2[c];
            s1 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s1), 0xB1, 0xF, 0xF, false));
            s2 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s2), 0xB1, 0xF, 0xF, false));
            s1 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s1), 0x4E, 0xF, 0xF, false));
            s2 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s2), 0x4E, 0xF, 0xF, false));
            s1 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s1), 0x141, 0xF, 0xF, false));
            s2 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s2), 0x141, 0xF, 0xF, false));
            s1 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s1), 0x140, 0xF, 0xF, false));
            s2 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s2), 0x140, 0xF, 0xF, false));
            if (fr == 0) {
                const int ch = (c < 8) ? 8 * fq + c : 32 + 8 * fq + (c - 8);
                red[wid][0][ch] = s1;
                red[wid][1][ch] = s2;
            }
        }
        __syncthreads();
        if (tid < 64) {
            const float s1 = red[0][0][tid] + red[1][0][tid] + red[2][0][tid] + red[3][0][tid];
            const float s2 = red[0][1][tid] + red[1][1][tid] + red[2][1][tid] + red[3][1][tid];
            float* row = a.bpart + (size_t)blockIdx.x * 2 * g.D1;
            row[tid] = s1;
            row[g.D1 + tid] = a.brstd[tid] * (s2 - a.bmean[tid] * s1);      // sum du * xhat
        }
    }
}

template <int T>
static void launch_thin_fwd2(const GG& g, const FwdArgs& a, int blocks, hipStream_t s) {
    const bool raw = a.y1 || a.y2;
    const int actm = !a.yact ? 0 : (a.eact == PAI_ACT_LRELU ? 1 : (a.eact == PAI_ACT_RELU ? 2 : 3));
#define TF2(RAWV, ACTV)                                                                                   \
    do {                                                                                                  \
        if (g.Cout == 64) PAI_LAUNCH((thin_fwd2_k<T, RAWV, ACTV, 1>), dim3(blocks), dim3(256), 0, s, g, a); \
        else PAI_LAUNCH((thin_fwd2_k<T, RAWV, ACTV, 2>), dim3(blocks), dim3(256), 0, s, g, a);             \
    } while (0)
    if (raw && a.bz) {      // thin_fwd_bwd_rows: T = 1, D1 = D2 = 64, no activated output
        PAI_LAUNCH((thin_fwd2_k<1, true, 0, 2, true>), dim3(blocks), dim3(256), 0, s, g, a);
    } else if (raw) {
        if (actm == 0) TF2(true, 0); else if (actm == 1) TF2(true, 1); else if (actm == 2) TF2(true, 2); else TF2(true, 3);
    } else {
        if (actm == 1) TF2(false, 1); else if (actm == 2) TF2(false, 2); else TF2(false, 3);
    }
#undef TF2
}

static int thin_fwd_blocks(const GG& g) {
    int blocks = cdiv(g.M, 64);
    static const int cap = getenv("PAI_TF_BLOCKS") ? atoi(getenv("PAI_TF_BLOCKS")) : 4096;
    return blocks > cap ? cap : blocks;
}

static bool thin_fwd2_ok(const GG& g, const FwdArgs& a) {
    const bool fast = pai_tunable("thin_fast", 1) && (g.W % 2) == 0 && (int64_t)g.N * g.H * g.W * 2 < (1ll << 31) &&
                      pai_tunable("thin_sect", 1) && (g.D2 == 0 || (g.D1 % 64) == 0);
    const bool out_ok = (a.y1 || a.y2 || a.yact) && (!a.yact || a.eact != PAI_ACT_TANH) && (!a.y2 || a.y1) &&
                        (int64_t)g.M * g.Cout * 2 < (1ll << 32) && (g.D2 == 0 || (g.D1 == 64 && g.D2 == 64));
    return pai_tunable("thin_fwd2", 1) && fast && g.ntaps == 16 && out_ok && (g.Cout == 64 || g.Cout == 128) &&
           !(a.yact && (a.y1 || a.y2) && g.D2 > 0);
}

// pai_conv_dgrad_bn on a thin -> wide input gradient: the producer's first BatchNorm-backward pass in the store
// (thin_fwd2_k<..., BWD>); returns the partial rows the launch writes (one per workgroup), 0 = not this kernel
int thin_fwd_bwd_rows(int dtype, const GG& g, const FwdArgs& a, int act1, const void* add, const float* scale) {
    if (!thin_fwd_ok(dtype, g, a) || !thin_fwd2_ok(g, a) || !pai_tunable("thin_bwd", 1)) return 0;
    if (g.C2 != 0 || g.D1 != 64 || g.D2 != 64 || !a.y1 || !a.y2 || a.yact || act1 != PAI_ACT_NONE || add || scale) return 0;
    return thin_fwd_blocks(g);
}

int launch_thin_fwd(const GG& g, const FwdArgs& a, hipStream_t s) {
    int blocks = thin_fwd_blocks(g);
    // thin_gather4: even row length, 32-bit byte offsets
    int fast_ok = pai_tunable("thin_fast", 1) && (g.W % 2) == 0 && (int64_t)g.N * g.H * g.W * 2 < (1ll << 31);
    // whole-sector stores need both 8-channel pieces of a lane in the same output tensor: 64-channel groups
    // (scripts/micro/convbench, thin_sect = 0 / 1: encoders[0] forward 43.7 -> 42.8 us, D block 0 forward 76.9 -> 75.2,
    // input gradient of decoders[7] 56.6 -> 54.0; bit-identical outputs)
    if (pai_tunable("thin_sect", 1) && (g.D2 == 0 || (g.D1 % 64) == 0)) fast_ok |= 2;
    // thin_fwd2_k: 4 x 4 kernels, whole-sector stores, every output below 4 GB, tanh not needed here, an output to write
    if (thin_fwd2_ok(g, a)) {
        if (g.C2 == 0) launch_thin_fwd2<1>(g, a, blocks, s);
        else launch_thin_fwd2<2>(g, a, blocks, s);
        PAI_LAUNCH_CHECK();
        return 0;
    }
    if (g.C2 == 0) PAI_LAUNCH(thin_fwd_k<1>, dim3(blocks), dim3(256), 0, s, g, a, fast_ok);
    else PAI_LAUNCH(thin_fwd_k<2>, dim3(blocks), dim3(256), 0, s, g, a, fast_ok);
    PAI_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// TH: thin -> wide with MANY output channels and a tiny 1-channel source: the input gradient of the PatchGAN head
// (Conv2d(512, 1, k4, s1, p1), reference models/wrapper.py:233): dx[n][y][x][c] = sum over the 16 taps of
// dl[n][y + dy_t][x + dx_t] * w[c][wt_t], times the activation derivative of the layer that produced the head's input
// (pai_conv_dgrad_act).  0.24 GFLOP against 33 MB written and 33 MB read at batch 128: a store stream.  It used to run
// on the generic vector-ALU kernel (51 us + a separate 15 us activation pass in the step); here a workgroup owns one
// image row (16 pixels x 512 channels), the source image and the transposed filter sit in LDS, and the derivative is
// applied in the store (round 4).
// ------------------------------------------------------------------------------------------------
bool head_dgrad_ok(int dtype, const GG& g, const FwdArgs& a) {
    return dtype == PAI_BF16 && pai_tunable("head_dgrad", 1) && g.nphase == 1 && g.ntaps == 16 && g.S == 1 && g.OS == 1 &&
           g.C1 == 1 && g.C2 == 0 && g.D2 == 0 && g.Cout == g.D1 && (g.Cout % 8) == 0 && g.Cout <= 1024 && g.OW <= 64 &&
           g.H * g.W <= 4096 && a.y1 && !a.y2 && !a.yact && !a.yf32 && !a.stats && !a.bias && !g.relu1 && !a.badd &&
           !a.bscale && !a.bpart && !a.skip_d1;
}

__global__ __launch_bounds__(256) void head_dgrad_k(GG g, FwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char hsm[];
    bf16_t* wt = (bf16_t*)hsm;                                   // [16 taps][Cout]
    float* src = (float*)(hsm + (size_t)16 * g.Cout * 2);        // the source image of this block's sample, H x W
    const int tid = threadIdx.x;
    const int n = blockIdx.x / g.OH, oy = blockIdx.x - n * g.OH;
    const bf16_t* w = (const bf16_t*)a.w;                        // [Cout][16] (one source channel)
    for (int i = tid; i < g.Cout * 16; i += 256) {
        const int c = i >> 4, t = i & 15;
        wt[t * g.Cout + c] = w[i];
    }
    const bf16_t* x = (const bf16_t*)a.x1 + (size_t)n * g.H * g.W;
    for (int i = tid; i < g.H * g.W; i += 256) src[i] = bf2f(x[i]);
    __syncthreads();
    const int groups = g.Cout / 8;                               // 8-channel pieces per pixel
    const bf16_t* az = (const bf16_t*)a.bz;
    for (int item = tid; item < g.OW * groups; item += 256) {
        const int ox = item / groups, cg = item - ox * groups;
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int sy = oy + g.dy[0][t], sx = ox + g.dx[0][t];
            if ((unsigned)sy >= (unsigned)g.H || (unsigned)sx >= (unsigned)g.W) continue;
            const float v = src[sy * g.W + sx];
            float wv[8];
            V8<bf16_t>::ld(wt + g.wt[0][t] * g.Cout + cg * 8, wv);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaf(v, wv[j], acc[j]);
        }
        const size_t o = ((size_t)(n * g.OH + oy) * g.OW + ox) * g.Cout + cg * 8;
        if (az) {          // pai_conv_dgrad_act: the bf16-rounded gradient times act'(stored activation)
            float zv[8];
            V8<bf16_t>::ld(az + o, zv);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float gq = bf2f(f2bf(acc[j]));
                acc[j] = a.bact1 == PAI_ACT_NONE ? gq : gq * act_grad(zv[j], a.bact1);
            }
        }
        V8<bf16_t>::st((bf16_t*)a.y1 + o, acc);
    }
}

int launch_head_dgrad(const GG& g, const FwdArgs& a, hipStream_t s) {
    const size_t lds = (size_t)16 * g.Cout * 2 + (size_t)g.H * g.W * 4;
    PAI_LAUNCH(head_dgrad_k, dim3(g.N * g.OH), dim3(256), lds, s, g, a);
    PAI_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// TD: wide -> thin (4-phase transposed form, Cout <= 2): skinny GEMM into fp32 scratch, then col2im.
// ------------------------------------------------------------------------------------------------
static int thin_dgrad_T(const GG& g, const FwdArgs& a) { return (g.Cout == 2 && !a.skip_d1) ? 2 : 1; }

int64_t thin_dgrad_scratch_bytes(const GG& g, const FwdArgs& a) {
    return (int64_t)g.N * g.H * g.W * thin_dgrad_T(g, a) * 16 * 4;
}

// stride-1 gathers with <= 2 output channels: the 4-phase transposed form (decoders[7], input
// gradient of D block 0) and the k4 s1 p1 PatchGAN head (models/wrapper.py:233)
bool thin_dgrad_shape_ok(int dtype, const GG& g) {
    const bool phases = g.nphase == 4 && g.ntaps == 4 && g.OS == 2;
    const bool conv1 = g.nphase == 1 && g.ntaps == 16 && g.OS == 1;
    // the 64 -> 1 3x3 out convolution of the residual / Trans U-Nets (models/res_unet.py:308, models/trans_unet.py:98)
    const bool conv3 = g.nphase == 1 && g.ntaps == 9 && g.OS == 1 && g.Cout == 1;
    const int ks = g.Cin / 32;
    return dtype == PAI_BF16 && g.S == 1 && (phases || conv1 || conv3) && g.Cout <= 2 && (g.C1 % 32) == 0 &&
           (g.C2 % 32) == 0 && (ks == 1 || ks == 2 || ks == 4 || ks == 8 || ks == 16);
}

bool thin_dgrad_ok(int dtype, const GG& g, const FwdArgs& a) {
    if (!thin_dgrad_shape_ok(dtype, g) || a.stats) return false;
    return pai_ctx()->scratch != nullptr && pai_ctx()->scratch_bytes >= thin_dgrad_scratch_bytes(g, a);
}

// KS = Cin / 32 is a template parameter: with a run-time trip count the fragment arrays are indexed
// dynamically and hipcc places them in scratch memory (528 B/lane, 10x slower)
template <int T, int KS>
__global__ __launch_bounds__(256) void thin_dgrad_gemm_k(GG g, FwdArgs a, float* Y, int t0) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const bf16_t* x1 = (const bf16_t*)a.x1;
    const bf16_t* x2 = (const bf16_t*)a.x2;
    const bf16_t* w = (const bf16_t*)a.w;
    const int Msrc = g.N * g.H * g.W; // one GEMM row per SOURCE pixel
    // A operand: Wp[(t, tap)][c]
    bf8_t af[T][KS];
#pragma unroll
    for (int tt = 0; tt < T; ++tt)
#pragma unroll
        for (int s = 0; s < KS; ++s)
            af[tt][s] = fr < g.wtaps ? *(const bf8_t*)(w + (size_t)((t0 + tt) * g.wtaps + fr) * g.Cin + 32 * s + 8 * fq)
                                     : __builtin_bit_cast(bf8_t, make_uint4(0, 0, 0, 0));   // 9-tap filters: rows 9..15 empty
    for (int p0 = (blockIdx.x * 4 + wid) * 16; p0 < Msrc; p0 += gridDim.x * 64) {
        const int m = min(p0 + fr, Msrc - 1);
        f4_t acc[T];
#pragma unroll
        for (int tt = 0; tt < T; ++tt) acc[tt] = (f4_t){0.f, 0.f, 0.f, 0.f};
        bf8_t bfr[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int c = 32 * s + 8 * fq;
            if (c < g.C1) {
                bfr[s] = *(const bf8_t*)(x1 + (size_t)m * g.C1 + c);
                if (g.relu1) bfr[s] = relu8(bfr[s]);
            } else {
                bfr[s] = *(const bf8_t*)(x2 + (size_t)m * g.C2 + (c - g.C1));
                if (g.relu2) bfr[s] = relu8(bfr[s]);
            }
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int tt = 0; tt < T; ++tt)
                acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[tt][s], bfr[s], acc[tt], 0, 0, 0);
        }
        if (p0 + fr < Msrc) {
#pragma unroll
            for (int tt = 0; tt < T; ++tt)
                *(float4*)(Y + ((size_t)(p0 + fr) * T + tt) * 16 + 4 * fq) =
                    make_float4(acc[tt][0], acc[tt][1], acc[tt][2], acc[tt][3]);
        }
    }
}

// One thread per output pixel.  32-bit index arithmetic (host: N * OH * OW < 2^31) with shifts where the extents are
// powers of two: the 64-bit divisions that stood here were most of the launch (decoders[7]: 36 -> see DESIGN.md).
// Lanes walk along ox, so the <= 4 source rows of 64 B a pixel reads are shared by its neighbours through L1.
__global__ __launch_bounds__(256) void thin_col2im_k(GG g, FwdArgs a, const float* Y, int T, int t0) {
    const unsigned total = (unsigned)(g.N * g.OH * g.OW);
    const bool phases = g.nphase == 4;
    // the phase differs from lane to lane: indexing the by-value tables with it means per-lane loads from the kernel
    // argument segment; a copy in LDS serves them in one cycle
    __shared__ int tab[4][16];     // dy | dx << 8 | wt << 16 per (phase, tap)
    if (threadIdx.x < 64) {
        const int p = threadIdx.x >> 4, k = threadIdx.x & 15;
        tab[p][k] = (g.dy[p][k] & 0xff) | ((g.dx[p][k] & 0xff) << 8) | ((g.wt[p][k] & 0xff) << 16);
    }
    __syncthreads();
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        unsigned ox, oy, n;
        if (g.ldw >= 0) {
            ox = i & ((unsigned)g.OW - 1u);
            oy = (i >> g.ldw) & ((unsigned)g.OH - 1u);
            n = i >> (g.ldw + g.ldh);
        } else {
            ox = i % (unsigned)g.OW;
            const unsigned r = i / (unsigned)g.OW;
            oy = r % (unsigned)g.OH;
            n = r / (unsigned)g.OH;
        }
        const int ph = phases ? (int)((oy & 1u) * 2u + (ox & 1u)) : 0;
        const int ay = (int)(phases ? oy >> 1 : oy), bx = (int)(phases ? ox >> 1 : ox);
        const unsigned rowbase = n * (unsigned)g.H;
        for (int tt = 0; tt < T; ++tt) {
            const int t = t0 + tt;
            float v = a.bias ? a.bias[t] : 0.f;
            for (int k = 0; k < g.ntaps; ++k) {
                const int e = tab[ph][k];
                const int iy = ay + (int)(signed char)(e & 0xff), ix = bx + (int)(signed char)((e >> 8) & 0xff);
                if ((unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W)
                    v += Y[(((size_t)((rowbase + (unsigned)iy) * (unsigned)g.W + (unsigned)ix)) * T + tt) * 16 + ((e >> 16) & 0xff)];
            }
            if (t < g.D1) {
                if (a.y1 && !a.skip_d1) ((bf16_t*)a.y1)[(size_t)i * g.D1 + t] = f2bf(v);
            } else if (a.y2) {
                ((bf16_t*)a.y2)[(size_t)i * g.D2 + (t - g.D1)] = f2bf(v);
            }
            if (a.yact || a.yf32) {
                const float av = act_apply(v, a.eact);
                if (a.yact) ((bf16_t*)a.yact)[(size_t)i * g.Cout + t] = f2bf(av);
                if (a.yf32) a.yf32[(size_t)i * g.Cout + t] = av;
            }
        }
    }
}

// Both steps in ONE kernel for the 4-phase form (decoders[7] + tanh, input gradient of D block 0): a workgroup owns a
// 16 x 16 block of SOURCE pixels of one image.  It runs the skinny GEMM over the block and its one-pixel halo (18 x 18
// pixels = 21 groups of 16, 1.27 x the block: the halo rows are L2 hits, the neighbouring workgroups read them too),
// keeps the 16 tap values per pixel in LDS (fp32, rows padded to 17 floats) and gathers the 32 x 32 output pixels from
// there -- the 67 MB (decoders[7]) of fp32 scratch written by thin_dgrad_gemm_k and re-read ~2 x by thin_col2im_k never
// exist.  Halo pixels beyond the image hold zeros (zero fragments), so the gather needs no bounds test.
constexpr int TU_HW = 18, TU_PIX = TU_HW * TU_HW, TU_GROUPS = (TU_PIX + 15) / 16, TU_LD = 17;
template <int T, int KS>
__global__ __launch_bounds__(256) void thin_up_k(GG g, FwdArgs a, int t0, int tiles_x, int tiles_y) {
    __shared__ float Ys[T][TU_GROUPS * 16][TU_LD];
    __shared__ int tab[4][4];     // dy | dx << 8 | wt << 16 per (phase, tap)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const bf16_t* x1 = (const bf16_t*)a.x1;
    const bf16_t* x2 = (const bf16_t*)a.x2;
    const bf16_t* w = (const bf16_t*)a.w;
    int bid = blockIdx.x;
    const int tx0 = (bid % tiles_x) * 16; bid /= tiles_x;
    const int ty0 = (bid % tiles_y) * 16;
    const int n = bid / tiles_y;
    if (tid < 16) {
        const int p = tid >> 2, k = tid & 3;
        tab[p][k] = (g.dy[p][k] & 0xff) | ((g.dx[p][k] & 0xff) << 8) | ((g.wt[p][k] & 0xff) << 16);
    }
    // A operand: Wp[(t, tap)][c]
    bf8_t af[T][KS];
#pragma unroll
    for (int tt = 0; tt < T; ++tt)
#pragma unroll
        for (int s = 0; s < KS; ++s)
            af[tt][s] = *(const bf8_t*)(w + (size_t)((t0 + tt) * g.wtaps + fr) * g.Cin + 32 * s + 8 * fq);
    // A wave's pixel groups (5 or 6 of the 21) are loaded TWO AT A TIME ahead of the MFMAs that consume them: with one
    // group per round trip the wave waited out an L2 / HBM latency five times over (round 4).
    auto load_group = [&](int gi, bf8_t (&bfr)[KS]) {
        const int p = gi * 16 + fr;
        const int hy = p / TU_HW, hx = p - hy * TU_HW;
        const int iy = ty0 - 1 + hy, ix = tx0 - 1 + hx;
        const bool inb = gi < TU_GROUPS && p < TU_PIX && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
        const size_t m = inb ? (size_t)(n * g.H + iy) * g.W + ix : 0;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int c = 32 * s + 8 * fq;
            bfr[s] = __builtin_bit_cast(bf8_t, make_uint4(0, 0, 0, 0));
            if (inb) {
                if (c < g.C1) bfr[s] = *(const bf8_t*)(x1 + m * g.C1 + c);
                else bfr[s] = *(const bf8_t*)(x2 + m * g.C2 + (c - g.C1));
            }
        }
    };
    bf8_t cur[KS], nxt[KS], nx2[KS];
    load_group(wid, cur);
    load_group(wid + 4, nxt);
    for (int gi = wid; gi < TU_GROUPS; gi += 4) {
        const int p = gi * 16 + fr;
        load_group(gi + 8, nx2);
        bf8_t bfr[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            bfr[s] = cur[s];
            if ((32 * s + 8 * fq < g.C1) ? g.relu1 : g.relu2) bfr[s] = relu8(bfr[s]);
            cur[s] = nxt[s];
            nxt[s] = nx2[s];
        }
#pragma unroll
        for (int tt = 0; tt < T; ++tt) {
            f4_t acc = (f4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[tt][s], bfr[s], acc, 0, 0, 0);
            // D[i = tap 4 fq + r][j = pixel fr]
#pragma unroll
            for (int r = 0; r < 4; ++r) Ys[tt][p][4 * fq + r] = acc[r];
        }
    }
    __syncthreads();
    // 32 x 32 output pixels, lanes along ox
    for (int i = tid; i < 1024; i += 256) {
        const int oyl = i >> 5, oxl = i & 31;
        const int oy = 2 * ty0 + oyl, ox = 2 * tx0 + oxl;
        const int ph = (oyl & 1) * 2 + (oxl & 1);
        const int hy0 = (oyl >> 1) + 1, hx0 = (oxl >> 1) + 1;
        const size_t o = ((size_t)n * g.OH + oy) * g.OW + ox;
#pragma unroll
        for (int tt = 0; tt < T; ++tt) {
            const int t = t0 + tt;
            float v = a.bias ? a.bias[t] : 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = tab[ph][k];
                const int hy = hy0 + (int)(signed char)(e & 0xff), hx = hx0 + (int)(signed char)((e >> 8) & 0xff);
                v += Ys[tt][hy * TU_HW + hx][(e >> 16) & 0xff];
            }
            if (t < g.D1) {
                if (a.y1 && !a.skip_d1) ((bf16_t*)a.y1)[o * g.D1 + t] = f2bf(v);
            } else if (a.y2) {
                ((bf16_t*)a.y2)[o * g.D2 + (t - g.D1)] = f2bf(v);
            }
            if (a.yact || a.yf32) {
                const float av = act_apply(v, a.eact);
                if (a.yact) ((bf16_t*)a.yact)[o * g.Cout + t] = f2bf(av);
                if (a.yf32) a.yf32[o * g.Cout + t] = av;
            }
        }
    }
}

static bool thin_up_ok(const GG& g) {
    return pai_tunable("thin_up", 1) && g.nphase == 4 && g.ntaps == 4 && g.OS == 2 && g.wtaps == 16 && (g.H % 16) == 0 &&
           (g.W % 16) == 0 && (g.Cin == 64 || g.Cin == 128);
}

int launch_thin_dgrad(const GG& g, const FwdArgs& a, hipStream_t s) {
    const int T = thin_dgrad_T(g, a);
    const int t0 = (g.Cout == 2 && a.skip_d1) ? 1 : 0;
    if (thin_up_ok(g)) {
        const int tx = g.W / 16, ty = g.H / 16;
        const dim3 grid(g.N * tx * ty);
        if (T == 1 && g.Cin == 128) PAI_LAUNCH((thin_up_k<1, 4>), grid, dim3(256), 0, s, g, a, t0, tx, ty);
        else if (T == 1) PAI_LAUNCH((thin_up_k<1, 2>), grid, dim3(256), 0, s, g, a, t0, tx, ty);
        else if (g.Cin == 128) PAI_LAUNCH((thin_up_k<2, 4>), grid, dim3(256), 0, s, g, a, t0, tx, ty);
        else PAI_LAUNCH((thin_up_k<2, 2>), grid, dim3(256), 0, s, g, a, t0, tx, ty);
        PAI_LAUNCH_CHECK();
        return 0;
    }
    // every wave keeps the whole filter in registers: give it >= 4 pixel groups to amortise that
    int blocks = cdiv((int64_t)g.N * g.H * g.W, 256);
    if (blocks > 4096) blocks = 4096;
#define TDG(TT, KK) PAI_LAUNCH((thin_dgrad_gemm_k<TT, KK>), dim3(blocks), dim3(256), 0, s, g, a, pai_ctx()->scratch, t0)
#define TDG_K(TT)                                     \
    switch (g.Cin / 32) {                             \
        case 1: TDG(TT, 1); break;                    \
        case 2: TDG(TT, 2); break;                    \
        case 4: TDG(TT, 4); break;                    \
        case 8: TDG(TT, 8); break;                    \
        case 16: TDG(TT, 16); break;                  \
        default: PAI_CHECK(false, "thin dgrad: unsupported Cin %d", g.Cin); \
    }
    if (T == 1) { TDG_K(1) } else { TDG_K(2) }
#undef TDG_K
#undef TDG
    PAI_LAUNCH_CHECK();
    PAI_CHECK((int64_t)g.N * g.OH * g.OW < (1ll << 31), "thin dgrad: more than 2^31 output pixels");
    int64_t b2 = ((int64_t)g.N * g.OH * g.OW + 255) / 256;
    if (b2 > 8192) b2 = 8192;
    PAI_LAUNCH(thin_col2im_k, dim3((int)b2), dim3(256), 0, s, g, a, (const float*)pai_ctx()->scratch, T, t0);
    PAI_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// TW: thin weight gradient.  wide tile (64 pixels x <= 128 channels) staged by LDS-DMA into the
// 256-B-row transposed-read image of the MFMA wgrad kernel; patch fragments gathered directly.
// (measured and dropped: two LDS stages with the next chunk's tile and gathers in flight during the MFMAs, one
//  barrier per chunk -- D block 0 at batch 128: 182 -> 216 us, at batch 64: 97 -> 113 us; like the patch-resident
//  weight gradient, this kernel hides its latency across workgroups and loses them to the second stage)
// ------------------------------------------------------------------------------------------------
struct ThinW {
    const bf16_t *thin1, *thin2;  // [N][TH][TW] one channel each; thin2 may be null (T = 1)
    const bf16_t *wide1, *wide2;  // [N][H][W][WC1] | [WC2]
    int N, H, W, TH, TW, WC1, WC2;
    int relu1, relu2;             // ReLU-on-load flags of the wide tensors
    int lw, lh;                   // log2 W, log2 H or -1
    int tmul, flip;               // thin pixel = tmul*wide + (flip ? 1 - k : k - 1)
    int kw, ntaps;                // tap grid: 4 x 4 (16) or 3 x 3 (9; patch rows 9..15 stay zero)
    float* dw;
    int s_wc, s_tap, s_t;         // dw index = wc*s_wc + tap*s_tap + t*s_t
    float* dbias;                 // per wide channel, or null
    int M;                        // N*H*W
    int fast;                     // patch rows by thin_gather8 (set by launch_tw)
    int pair;                     // 64 wide channels: two 64-pixel chunks share one 256-B-row LDS tile (set by launch_tw)
    float* partial;               // per-workgroup partial sums [grid.x][grid.y][16 T + 1][128] or null (atomics)
};

__device__ __forceinline__ int tw_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ unsigned tw_off(int row, int ch) { return (unsigned)(256 * row + 16 * (ch ^ tw_swz(row))); }

#ifndef THIN_WG_WAVES
#define THIN_WG_WAVES 1
#endif
template <int T>
__global__ __launch_bounds__(256, THIN_WG_WAVES) void thin_wgrad_k(ThinW p, int chunks_per_block) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // 64 x 256 B
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4, tq = fr >> 2, tp = fr & 3;
    const int wc0 = blockIdx.y * 128;          // channel group of this workgroup (WC > 128: single source)
    const int WC = min(p.WC1 + p.WC2 - wc0, 128);
    // pair (64 wide channels, one source): the 256-B rows of the LDS tile would be half empty -- instead row r holds
    // pixel r of TWO consecutive 64-pixel chunks side by side (128 B each), the waves' column halves become the two
    // chunks, and the step covers 128 pixels: no zero-line fills, half the barriers and the patch gathers of a wave
    // serve 64 instead of 32 channels.  Columns 64..127 of the partial tile fold onto 0..63 in the reduction.
    const int pair = p.pair;
    const int ntile = pair ? 8 : WC / 16;      // column tiles of 16 wide channels (4 or 8)
    const int ks = wid & 1;                    // this wave's 32-pixel K step of the 64-pixel chunk
    const int nt0 = (wid >> 1) * (ntile / 2);  // and its half of the column tiles
    const int ntn = ntile / 2;
    const int pixh = pair ? (wid >> 1) * 64 : 0;   // pair: first pixel of this wave's chunk within the step
    const int CH = pair ? 128 : 64;            // pixels per step
    const bf16_t* zero = (const bf16_t*)g_zero_line_thin;

    // LDS-DMA map: position (row sr + 16j, slot sc) holds logical chunk sc ^ swz(row)
    const int sc = lane & 15, sr = wid * 4 + (lane >> 4);
    const int gch = sc ^ tw_swz(sr);
    const bf16_t* wsrc;
    int wstride, wcol;
    bool wvalid = pair || gch * 8 < WC;
    const int wpix = pair ? (gch >> 3) * 64 : 0;   // pair: slots 8..15 of a row belong to the second chunk
    if (pair) { wsrc = p.wide1; wstride = p.WC1; wcol = (gch & 7) * 8; }
    else if (wc0 + gch * 8 < p.WC1) { wsrc = p.wide1; wstride = p.WC1; wcol = wc0 + gch * 8; }
    else { wsrc = p.wide2; wstride = p.WC2; wcol = gch * 8 - p.WC1; }

    // patch row k = 16*tt + fr -> (tap, t)
    int kdy[T], kdx[T];
    const bf16_t* ksrc[T];
    int kt[T];
#pragma unroll
    for (int tt = 0; tt < T; ++tt) {
        const int k = 16 * tt + fr;
        const int tap = k / T, t = k - tap * T;
        const int th = tap / p.kw, tw = tap - th * p.kw;
        kdy[tt] = p.flip ? 1 - th : th - 1;
        kdx[tt] = tap < p.ntaps ? (p.flip ? 1 - tw : tw - 1) : -(1 << 20);   // rows beyond the tap count: never in bounds
        ksrc[tt] = t ? p.thin2 : p.thin1;
        kt[tt] = t;
    }
    // fast patch rows: the 8 pixels of a lane lie in one image row (W % 8 == 0) and the thin tensors fit 32-bit byte
    // offsets -> thin_gather8; host sets p.fast
    const unsigned thin_bytes = (unsigned)(p.N * p.TH * p.TW) * 2u;
    const __amdgpu_buffer_rsrc_t trs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.thin1), 0, thin_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t trs2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(p.thin2 ? p.thin2 : p.thin1), 0, p.thin2 ? thin_bytes : 0u, 0x00020000);

    f4_t acc[T][4];
#pragma unroll
    for (int tt = 0; tt < T; ++tt)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[tt][j] = (f4_t){0.f, 0.f, 0.f, 0.f};
    // bias gradient = column sums of the wide tile: one more MFMA per column tile with a fragment whose row 0 is all ones
    // (D row 0 = sum over the 32 pixels of the K step).  The 32 two-byte LDS reads per thread and chunk that stood here
    // were more LDS instructions than the rest of the step.
    f4_t accb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) accb[j] = (f4_t){0.f, 0.f, 0.f, 0.f};
    const unsigned one2 = fr == 0 ? 0x3f803f80u : 0u;      // bf16 1.0 pairs in MFMA row 0
    const bf8_t ones = __builtin_bit_cast(bf8_t, make_uint4(one2, one2, one2, one2));

    for (int ci = 0; ci < chunks_per_block; ++ci) {
        const int p0 = (blockIdx.x * chunks_per_block + ci) * CH;
        if (p0 >= p.M) break;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = p0 + wpix + sr + 16 * j;
            const bf16_t* src = (wvalid && m < p.M) ? wsrc + (size_t)m * wstride + wcol : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(smem + (16 * j + wid * 4) * 256),
                                             16, 0, 0);
        }
        // A operand: patch[pix = p0 + 32*ks + 8*fq + j][k]
        bf8_t af[T];
#pragma unroll
        for (int tt = 0; tt < T; ++tt) {
            us8_t pv;
            if (p.fast) {
                const int m = p0 + pixh + 32 * ks + 8 * fq;     // first of the lane's 8 pixels, all in one image row
                int n, ay, bx;
                if (p.lw >= 0) { bx = m & (p.W - 1); ay = (m >> p.lw) & (p.H - 1); n = m >> (p.lw + p.lh); }
                else { bx = m % p.W; const int r = m / p.W; ay = r % p.H; n = r / p.H; }
                const int iy = p.tmul * ay + kdy[tt], ix0 = p.tmul * bx + kdx[tt];
                const bool rowok = m < p.M && (unsigned)iy < (unsigned)p.TH && kdx[tt] > -(1 << 19);
                const int e0 = (n * p.TH + iy) * p.TW;
                if (p.tmul == 2) pv = kt[tt] ? thin_gather8<2>(trs2, e0, ix0, p.TW, rowok) : thin_gather8<2>(trs1, e0, ix0, p.TW, rowok);
                else pv = kt[tt] ? thin_gather8<1>(trs2, e0, ix0, p.TW, rowok) : thin_gather8<1>(trs1, e0, ix0, p.TW, rowok);
            } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int m = p0 + pixh + 32 * ks + 8 * fq + j;
                int n, ay, bx;
                if (p.lw >= 0) { bx = m & (p.W - 1); ay = (m >> p.lw) & (p.H - 1); n = m >> (p.lw + p.lh); }
                else { bx = m % p.W; const int r = m / p.W; ay = r % p.H; n = r / p.H; }
                const int iy = p.tmul * ay + kdy[tt], ix = p.tmul * bx + kdx[tt];
                const bool inb = m < p.M && (unsigned)iy < (unsigned)p.TH && (unsigned)ix < (unsigned)p.TW;
                pv[j] = inb ? ksrc[tt][(size_t)(n * p.TH + iy) * p.TW + ix] : (unsigned short)0;
            }
            }
            af[tt] = __builtin_bit_cast(bf8_t, pv);
        }
        __syncthreads();
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            if (nt >= ntn) break;
            const int col0 = (nt0 + nt) * 16;
            bf8_t bfr;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int row = ks * 32 + fq * 8 + h * 4 + tq;
                const int ch = col0 / 8 + (tp >> 1);
                bf4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                    (bf4_t __attribute__((address_space(3)))*)(smem + tw_off(row, ch) + 8 * (tp & 1)));
#pragma unroll
                for (int e = 0; e < 4; ++e) bfr[h * 4 + e] = v[e];
            }
            if ((pair || wc0 + col0 < p.WC1) ? p.relu1 : p.relu2) bfr = relu8(bfr);
#pragma unroll
            for (int tt = 0; tt < T; ++tt)
                acc[tt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[tt], bfr, acc[tt][nt], 0, 0, 0);
            if (p.dbias) accb[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, bfr, accb[nt], 0, 0, 0);
        }
        __syncthreads();
    }
    // D[k = 16*tt + 4*fq + r][wc = 16*(nt0+nt) + fr]
    // two partial tiles per workgroup: the two 32-pixel halves (ks) of a chunk are separate waves
    float* part0 = p.partial ? p.partial + (((size_t)blockIdx.x * 2) * gridDim.y + blockIdx.y) * ((16 * T + 1) * 128) : nullptr;
    float* part = part0 ? part0 + (size_t)ks * gridDim.y * ((16 * T + 1) * 128) : nullptr;
#pragma unroll
    for (int tt = 0; tt < T; ++tt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            if (nt >= ntn) break;
            const int wcl = (nt0 + nt) * 16 + fr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = 16 * tt + 4 * fq + r;
                if (part) {
                    part[k * 128 + wcl] = acc[tt][nt][r];
                } else {
                    const int tap = k / T, t = k - tap * T;
                    if (tap < p.ntaps)
                        atomicAdd(p.dw + (size_t)(wc0 + (pair ? (wcl & 63) : wcl)) * p.s_wc + tap * p.s_tap + t * p.s_t, acc[tt][nt][r]);
                }
            }
        }
    if (p.dbias) {
        // D row 0 (fq = 0, r = 0) of column tile nt0 + nt: lane fr holds column 16 (nt0 + nt) + fr of this wave's K-step half
        float* red = (float*)smem;
        __syncthreads();
        if (fq == 0) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                if (nt < ntn) red[ks * 128 + (nt0 + nt) * 16 + fr] = accb[nt][0];
        }
        __syncthreads();
        if (tid < (pair ? 128 : WC)) {
            if (part0) {
                part0[16 * T * 128 + tid] = red[tid] + red[tid + 128];
                part0[(size_t)gridDim.y * ((16 * T + 1) * 128) + 16 * T * 128 + tid] = 0.f;
            } else {
                atomicAdd(p.dbias + wc0 + (pair ? (tid & 63) : tid), red[tid] + red[tid + 128]);
            }
        }
    }
}

// Second stage of the weight gradient: dw += sum over workgroups of their partial tiles (16 atomics per
// element instead of one per workgroup and element -- 512 workgroups x 1024 atomics on 1024 addresses
// cost ~100 us).  grid.y = 16 ranges of tiles; thread = (element, 1 of 4 slices of the range).
__global__ __launch_bounds__(256) void thin_wgrad_reduce_k(ThinW p, int T, int ntiles, int groups) {
    __shared__ float red[4][64];
    const int pst = (16 * T + 1) * 128;
    const int e = blockIdx.x * 64 + (threadIdx.x & 63), slice = threadIdx.x >> 6;
    const int gy = e / pst, le = e - gy * pst;
    const int per = (ntiles + gridDim.y - 1) / gridDim.y;
    const int b0 = blockIdx.y * per, b1 = min(ntiles, b0 + per);
    float sum = 0.f;
    if (gy < groups) {
        // eight independent loads in flight per thread: as a plain loop this was a chain of ~32 dependent memory round
        // trips per thread and the launch took 21 us for 18 MB
        const float* src = p.partial + (size_t)gy * pst + le;
        const size_t bstride = (size_t)groups * pst;
        int b = b0 + slice;
        for (; b + 28 < b1; b += 32) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = src[(size_t)(b + 4 * u) * bstride];
            sum += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
        }
        for (; b < b1; b += 4) sum += src[(size_t)b * bstride];
    }
    red[slice][threadIdx.x & 63] = sum;
    __syncthreads();
    if (slice == 0 && gy < groups) {
        sum = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        const int k = le >> 7;
        int wc = gy * 128 + (le & 127);
        if (p.pair) wc &= 63;          // the second chunk's columns fold onto the channels
        if (wc < p.WC1 + p.WC2) {
            if (k < 16 * T) {
                const int tap = k / T, t = k - tap * T;
                if (tap < p.ntaps) atomicAdd(p.dw + (size_t)wc * p.s_wc + tap * p.s_tap + t * p.s_t, sum);
            } else if (p.dbias) {
                atomicAdd(p.dbias + wc, sum);
            }
        }
    }
}

__global__ __launch_bounds__(256) void sum1_k(const bf16_t* x, int64_t n, float* out) {
    __shared__ float ws[4];
    float s = 0.f;
    // 16-B loads (x is a device allocation: 16-B aligned), the tail element by element
    const int64_t n8 = n >> 3;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const uint4 v = ((const uint4*)x)[i];
        const unsigned w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) s += __uint_as_float(w4[e] << 16) + __uint_as_float(w4[e] & 0xffff0000u);
    }
    for (int64_t i = n8 * 8 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += bf2f(x[i]);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, ws[0] + ws[1] + ws[2] + ws[3]);
}

static int ilog2_exact(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return (1 << l) == v ? l : -1;
}

// conv-form thin wgrad (encoders[0], D block 0): thin = layer input, wide = dy
bool thin_wgrad_conv_ok(int dtype, const GG& g) {
    return dtype == PAI_BF16 && g.nphase == 1 && g.ntaps == 16 && g.S == 2 && g.OS == 1 && g.C1 == 1 &&
           g.C2 <= 1 && (g.Cout % 32) == 0 && g.Cout <= 128 && !g.relu1 && !g.relu2;
}
// transposed thin wgrad (decoders[7]): thin = dy (1 channel), wide = layer input x1|x2
bool thin_wgrad_convt_ok(int dtype, const GG& g) {
    return dtype == PAI_BF16 && g.nphase == 4 && g.ntaps == 4 && g.Cout == 1 && (g.C1 % 8) == 0 &&
           (g.C2 % 8) == 0 && (g.Cin % 32) == 0 && g.Cin <= 128;
}

// workgroups along the pixel range; with the two-stage reduction more of them cost nothing
static int tw_blocks(int64_t M, int groups, bool two_stage) {
    static const int cap_env = getenv("PAI_TW_BLOCKS") ? atoi(getenv("PAI_TW_BLOCKS")) : 0;
    int cap = cap_env ? cap_env : (two_stage ? 1024 / groups : 512);   // ~4 workgroups per CU over all channel groups
    if (cap < 64) cap = 64;
    const int chunks = cdiv(M, 64);
    int blocks = chunks < cap ? chunks : cap;
    const int cpb = cdiv(chunks, blocks);
    return cdiv(chunks, cpb);
}

// bytes of registered scratch the two-stage path of a thin weight gradient wants (its TAIL is used, so that
// a thin input gradient running on another stream can use the head at the same time)
int64_t thin_wgrad_scratch_bytes(int64_t M, int T, int WC) {
    return (int64_t)tw_blocks(M, cdiv(WC, 128), true) * 2 * cdiv(WC, 128) * (16 * T + 1) * 128 * sizeof(float);
}

static int launch_tw(ThinW& p, int T, hipStream_t s) {
    p.M = p.N * p.H * p.W;
    p.fast = pai_tunable("thin_fast", 1) && (p.W % 8) == 0 && (p.TW % 2) == 0 && (p.tmul == 1 || p.tmul == 2) &&
             (int64_t)p.N * p.TH * p.TW * 2 < (1ll << 31);
    p.lw = ilog2_exact(p.W);
    p.lh = ilog2_exact(p.H);
    if (p.lw < 0 || p.lh < 0) p.lw = p.lh = -1;
    // measured (scripts/micro/convbench): D block 0 (two thin channels, 2.1 M pixels) 127 -> 98 us, encoders[0] (one thin
    // channel) 45 -> 48 us -- the pairing pays where the patch gathers are the larger half of the step
    p.pair = pai_tunable("thin_pair", 1) && p.WC1 == 64 && p.WC2 == 0 && T == 2;
    const int chunks = cdiv(p.M, p.pair ? 128 : 64);
    const int groups = cdiv(p.WC1 + p.WC2, 128);
    const int64_t need = thin_wgrad_scratch_bytes(p.M, T, p.WC1 + p.WC2);
    static const bool no_two = getenv("PAI_TW_ATOMIC") && atoi(getenv("PAI_TW_ATOMIC")) != 0;
    const pai_handle_s* ctx = pai_ctx();
    const bool two_stage = !no_two && ctx->scratch != nullptr && ctx->scratch_bytes >= need;
    int blocks = tw_blocks(p.M, groups, two_stage);
    const int cpb = cdiv(chunks, blocks);
    p.partial = two_stage ? (float*)((char*)ctx->scratch + (ctx->scratch_bytes - need)) : nullptr;
    if (T == 1) PAI_LAUNCH(thin_wgrad_k<1>, dim3(blocks, groups), dim3(256), 64 * 256, s, p, cpb);
    else PAI_LAUNCH(thin_wgrad_k<2>, dim3(blocks, groups), dim3(256), 64 * 256, s, p, cpb);
    PAI_LAUNCH_CHECK();
    if (two_stage) {
        const int elems = groups * (16 * T + 1) * 128;
        PAI_LAUNCH(thin_wgrad_reduce_k, dim3(cdiv(elems, 64), 16), dim3(256), 0, s, p, T, blocks * 2, groups);
        PAI_LAUNCH_CHECK();
    }
    return 0;
}

int launch_thin_wgrad_conv(const GG& g, const WgradArgs& a, hipStream_t s) {
    ThinW p;
    memset(&p, 0, sizeof(p));
    const int T = g.C1 + g.C2;
    p.thin1 = (const bf16_t*)a.x1; p.thin2 = (const bf16_t*)a.x2;
    p.wide1 = (const bf16_t*)a.dy; p.wide2 = nullptr;
    p.N = g.N; p.H = g.OHg; p.W = g.OWg; p.TH = g.H; p.TW = g.W;
    p.WC1 = g.Cout; p.WC2 = 0;
    p.dw = a.dw; p.s_wc = 16 * T; p.s_tap = T; p.s_t = 1;   // fwd pack [Cout][16][T]
    p.dbias = a.dbias;
    p.tmul = 2; p.flip = 0; p.kw = 4; p.ntaps = 16;
    return launch_tw(p, T, s);
}

int launch_thin_wgrad_convt(const GG& g, const WgradArgs& a, hipStream_t s) {
    ThinW p;
    memset(&p, 0, sizeof(p));
    p.thin1 = (const bf16_t*)a.dy; p.thin2 = nullptr;
    p.wide1 = (const bf16_t*)a.x1; p.wide2 = (const bf16_t*)a.x2;
    p.N = g.N; p.H = g.H; p.W = g.W; p.TH = g.OH; p.TW = g.OW;
    p.WC1 = g.C1; p.WC2 = g.C2; p.relu1 = g.relu1; p.relu2 = g.relu2;
    p.dw = a.dw; p.s_wc = 1; p.s_tap = g.Cin; p.s_t = 0;     // fwd pack [1][16][Cin]
    p.dbias = nullptr;
    p.tmul = 2; p.flip = 0; p.kw = 4; p.ntaps = 16;
    if (launch_tw(p, 1, s)) return 1;
    if (a.dbias) {
        const int64_t n = (int64_t)g.N * g.OH * g.OW;
        PAI_LAUNCH(sum1_k, dim3(256), dim3(256), 0, s, (const bf16_t*)a.dy, n, a.dbias);
        PAI_LAUNCH_CHECK();
    }
    return 0;
}

// k4 s1 p1 conv with one output channel (PatchGAN head): thin = dy, wide = layer input;
// dW[kh][kw][c] = sum_pix x[a][b][c] * dy[a + 1 - kh][b + 1 - kw]
bool thin_wgrad_conv1_ok(int dtype, const GG& g) {
    return dtype == PAI_BF16 && g.nphase == 1 && g.ntaps == 16 && g.S == 1 && g.OS == 1 && g.Cout == 1 &&
           g.C2 == 0 && (g.C1 % 128) == 0;
}

int launch_thin_wgrad_conv1(const GG& g, const WgradArgs& a, hipStream_t s) {
    ThinW p;
    memset(&p, 0, sizeof(p));
    p.thin1 = (const bf16_t*)a.dy; p.thin2 = nullptr;
    p.wide1 = (const bf16_t*)a.x1; p.wide2 = nullptr;
    p.N = g.N; p.H = g.H; p.W = g.W; p.TH = g.OH; p.TW = g.OW;
    p.WC1 = g.C1; p.WC2 = 0; p.relu1 = g.relu1;
    p.dw = a.dw; p.s_wc = 1; p.s_tap = g.Cin; p.s_t = 0;     // fwd pack [1][16][Cin]
    p.dbias = nullptr;
    p.tmul = 1; p.flip = 1; p.kw = 4; p.ntaps = 16;
    if (launch_tw(p, 1, s)) return 1;
    if (a.dbias) {
        const int64_t n = (int64_t)g.N * g.OH * g.OW;
        PAI_LAUNCH(sum1_k, dim3(64), dim3(256), 0, s, (const bf16_t*)a.dy, n, a.dbias);
        PAI_LAUNCH_CHECK();
    }
    return 0;
}

// 3x3 "same" convolutions with one thin side (models/res_unet.py:265,308, models/trans_unet.py:66,98) on the same kernel:
// in_conv (1 -> 64): thin = layer input, wide = dy, dW[co][kh][kw] = sum_pix dy[a][b][co] * x[a + kh - 1][b + kw - 1]
bool thin_wgrad_conv3_ok(int dtype, const GG& g) {
    return dtype == PAI_BF16 && g.nphase == 1 && g.ntaps == 9 && g.S == 1 && g.OS == 1 && g.C1 == 1 && g.C2 == 0 &&
           (g.Cout % 32) == 0 && g.Cout <= 128 && !g.relu1;
}

int launch_thin_wgrad_conv3(const GG& g, const WgradArgs& a, hipStream_t s) {
    ThinW p;
    memset(&p, 0, sizeof(p));
    p.thin1 = (const bf16_t*)a.x1; p.thin2 = nullptr;
    p.wide1 = (const bf16_t*)a.dy; p.wide2 = nullptr;
    p.N = g.N; p.H = g.OHg; p.W = g.OWg; p.TH = g.H; p.TW = g.W;
    p.WC1 = g.Cout; p.WC2 = 0;
    p.dw = a.dw; p.s_wc = 9; p.s_tap = 1; p.s_t = 0;        // fwd pack [Cout][9][1]
    p.dbias = a.dbias;
    p.tmul = 1; p.flip = 0; p.kw = 3; p.ntaps = 9;
    return launch_tw(p, 1, s);
}

// out convolution (64 -> 1): thin = dy, wide = layer input, dW[kh][kw][c] = sum_pix x[a][b][c] * dy[a + 1 - kh][b + 1 - kw]
bool thin_wgrad_conv3t_ok(int dtype, const GG& g) {
    return dtype == PAI_BF16 && g.nphase == 1 && g.ntaps == 9 && g.S == 1 && g.OS == 1 && g.Cout == 1 && g.C2 == 0 &&
           (g.C1 % 32) == 0 && g.C1 <= 128;
}

int launch_thin_wgrad_conv3t(const GG& g, const WgradArgs& a, hipStream_t s) {
    ThinW p;
    memset(&p, 0, sizeof(p));
    p.thin1 = (const bf16_t*)a.dy; p.thin2 = nullptr;
    p.wide1 = (const bf16_t*)a.x1; p.wide2 = nullptr;
    p.N = g.N; p.H = g.H; p.W = g.W; p.TH = g.OH; p.TW = g.OW;
    p.WC1 = g.C1; p.WC2 = 0; p.relu1 = g.relu1;
    p.dw = a.dw; p.s_wc = 1; p.s_tap = g.Cin; p.s_t = 0;     // fwd pack [1][9][Cin]
    p.dbias = nullptr;
    p.tmul = 1; p.flip = 1; p.kw = 3; p.ntaps = 9;
    if (launch_tw(p, 1, s)) return 1;
    if (a.dbias) {
        const int64_t n = (int64_t)g.N * g.OH * g.OW;
        PAI_LAUNCH(sum1_k, dim3(64), dim3(256), 0, s, (const bf16_t*)a.dy, n, a.dbias);
        PAI_LAUNCH_CHECK();
    }
    return 0;
}

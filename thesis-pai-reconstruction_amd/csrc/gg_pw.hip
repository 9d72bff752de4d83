// Streaming pointwise (1 x 1) convolution: forward and input gradient of the bottleneck / expansion / skip convolutions of
// the residual blocks at the fine levels (reference models/res_unet.py:143-147, 74: ResidualBlockNeXt 1x1 -> grouped 3x3 ->
// 1x1, conv_skip), bf16.
//
//   out[M][COUT] = in[M][CIN] x W[COUT][CIN]^T (+ bias)        CIN x COUT <= 128 x 128 per wave (128 x 256 / 256 x 128: two
//                                                              waves share a pixel stream and split the output channels)
//
// At 512 x 512 x 16 images these layers move 1.6 GB for 69-137 GFLOP: HBM-bound by 5-10x.  The 128 x 128 tile kernel
// (gg_fwd_mfma_k: operands through LDS, a staged epilogue, one tile per workgroup) ran them at 2.2-2.6 TB/s; this one
// follows pw_k (the attention gate's kernel, gg_mfma.hip) and the streaming rules of ew_stream.hip:
//   * no LDS for the operands: a wave keeps the WHOLE filter in registers as the MFMA's A operand (rows permuted so that a
//     lane ends up with 8-channel chunks of one pixel and a store instruction writes 64 contiguous bytes per pixel) and
//     streams groups of 16 pixels as the B operand -- one 16-byte load per lane and 32 input channels;
//   * T groups per batch and the NEXT batch's loads issued before the current batch's MFMAs: 16-64 KB in flight per CU;
//   * the input may be two tensors read as one concatenation (decoder blocks), the output may split into two (their input
//     gradients);
//   * bias; BatchNorm partial statistics (one row per workgroup: 256 or 4096 rows instead of one per 128 pixels); the input
//     read through the producer's BatchNorm + ReLU (pai_conv_fwd_pro); the producer's BatchNorm backward, first pass, in
//     the store of an input gradient (pai_conv_dgrad_bn).
#include "common.h"
#include "gg_tile.h"

namespace {
constexpr int PWX_MAX_BLOCKS = 4096;

__device__ __forceinline__ float dpp_add16(float u) {      // sum over the 16 lanes of a row (all lanes get the total)
    u += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, u), 0xB1, 0xF, 0xF, false));
    u += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, u), 0x4E, 0xF, 0xF, false));
    u += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, u), 0x141, 0xF, 0xF, false));
    u += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, u), 0x140, 0xF, 0xF, false));
    return u;
}
}  // namespace

// the instantiations of launch_pwx: the filter fits the registers of a wave (CIN x COUT <= 128 x 128)
bool pwx_shape_ok(int cin, int cout) {
    return (cin == 64 && (cout == 64 || cout == 128 || cout == 256)) || (cin == 128 && (cout == 64 || cout == 128)) ||
           (cin == 256 && cout == 64) ||
           // two waves of a workgroup share a pixel stream and take half of the output channels each
           (pai_tunable("pwx_split", 1) && ((cin == 128 && cout == 256) || (cin == 256 && cout == 128)));
}

bool pwx_ok(int dtype, const GG& g, const FwdArgs& a) {
    if (dtype != PAI_BF16 || g.ntaps != 1 || g.nphase != 1 || g.S != 1 || g.OS != 1 || g.gslice) return false;
    if (!pwx_shape_ok(g.Cin, g.Cout) || (g.C1 % 32) || (g.C2 % 32)) return false;
    if (g.D2 ? ((g.D1 % 8) || (g.D2 % 8) || g.D1 + g.D2 != g.Cout) : false) return false;
    if (g.relu1 || g.relu2 || a.yf32 || a.skip_d1) return false;
    // fused producer backward (pai_conv_dgrad_bn): du = act1'(bz * bscale + bshift) * dgrad with the BatchNorm partial sums;
    // one destination, no second gradient, ReLU or none, no forward statistics / prologue at the same time
    if (a.bz && (g.D2 || a.badd || !a.bscale || !a.bshift || a.stats || a.pscale ||
                 (a.bpart && !(a.bmean && a.brstd)) || (a.bact1 != PAI_ACT_NONE && a.bact1 != PAI_ACT_RELU)))
        return false;
    if (a.pscale && (g.C2 || !a.pshift || (a.pact != PAI_ACT_NONE && a.pact != PAI_ACT_RELU))) return false;
    if (!a.y1 || a.yact) return false;                              // the raw output only (what a BatchNorm or a sum follows)
    if ((int64_t)g.M < 16384) return false;                         // small images: the tile kernels (split-K) do better
    return pai_tunable("pwx", 1) != 0;
}

int pwx_rows(const GG& g) {
    const int64_t b = ((int64_t)g.M + 255) / 256;
    // one resident workgroup per CU walks its share of a mid-size tensor (16 x 256 x 256: 116 against 129 us for 128 -> 128
    // channels, 32 against 43 us for 256 -> 64 at 128 x 128); the 4 M-pixel layers do better in many short workgroups
    // (299 against 315 us) -- scripts/bench_pw.py pwx_blocks=...
    const int64_t cap = pai_tunable("pwx_blocks", g.M >= (1 << 21) ? PWX_MAX_BLOCKS : 256);
    return (int)(b > cap ? cap : (b < 1 ? 1 : b));
}

// MODE 0: plain; 1: BatchNorm partial statistics of the output (forward); 2: the producer's backward in the store
// (input gradient: du = act'(bz * bscale + bshift) * dgrad, partial sums of du and du * (bz - bmean))
// NSPLIT (1 or 2): the waves (wid % NSPLIT) of a workgroup take COUT output channels each of a layer with NSPLIT * COUT, on the
// same pixel groups (the second wave's reads of x are L1 / L2 hits): 128 -> 256 and 256 -> 128 channels
template <int CIN, int COUT, int T, int MODE, bool PRE, int NSPLIT = 1>
__global__ __launch_bounds__(256) void pwx_k(GG g, FwdArgs a, int groups_per_wave) {
    constexpr bool STATS = MODE != 0;
    constexpr int KB = CIN / 32, NTT = COUT / 16, CL = COUT / 4, NCH = CL / 8, CALL = COUT * NSPLIT;
    __shared__ __attribute__((aligned(16))) float sbias[CALL];
    __shared__ float sred[4][2][STATS ? COUT : 1];
    __shared__ __attribute__((aligned(16))) float sbwd[3][MODE == 2 ? CALL : 4];      // bscale | bshift | bmean
    __shared__ __attribute__((aligned(16))) float spre[2][PRE ? CIN : 4];      // prologue: scale | shift per input channel
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int cbase = (wid % NSPLIT) * COUT;        // this wave's first output channel
    const int pw = wid / NSPLIT;                    // its pixel stream (4 / NSPLIT per workgroup)
    const bf16_t* x1 = (const bf16_t*)a.x1;
    const bf16_t* x2 = (const bf16_t*)a.x2;
    const bf16_t* w = (const bf16_t*)a.w + (size_t)cbase * CIN;
    const int C1 = g.C1, C2 = g.C2;
    for (int c = tid; c < CALL; c += 256) sbias[c] = a.bias ? a.bias[c] : 0.f;
    if (PRE) {
        for (int c = tid; c < CIN; c += 256) { spre[0][c] = a.pscale[c]; spre[1][c] = a.pshift[c]; }
    }
    const float plo = a.pact == PAI_ACT_RELU ? 0.f : -INFINITY;      // prologue activation: ReLU or none
    if (MODE == 2) {
        for (int c = tid; c < CALL; c += 256) {
            sbwd[0][c] = a.bscale[c];
            sbwd[1][c] = a.bshift[c];
            sbwd[2][c] = a.bmean ? a.bmean[c] : 0.f;
        }
    }
    const bf16_t* bzp = (const bf16_t*)a.bz;
    const bool brelu = a.bact1 == PAI_ACT_RELU;
    // filter: MFMA row (nt, i = fr) carries output channel 32 (nt >> 1) + 8 (i >> 2) + 4 (nt & 1) + (i & 3): lane (fr, fq)
    // ends up with channels 32 h + 8 fq + [0, 8) of pixel fr for h = 0 .. NCH - 1, so that ONE store instruction (fixed h)
    // writes 64 contiguous bytes per pixel (CL fq + 8 h, pw_k's order, gives 16-byte pieces 64 bytes apart)
    bf8_t wf[NTT][KB];
#pragma unroll
    for (int nt = 0; nt < NTT; ++nt)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
            wf[nt][kb] = *(const bf8_t*)(w + (size_t)(32 * (nt >> 1) + 8 * (fr >> 2) + 4 * (nt & 1) + (fr & 3)) * CIN + kb * 32 + fq * 8);
    // destination of this lane's 8-channel chunks (the raw output, split D1 | D2 for the input gradient of a concatenation)
    bf16_t* dst[NCH];
    int dstride[NCH];
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
        const int ch = cbase + 32 * h + 8 * fq;
        if (g.D2 && ch >= g.D1) { dst[h] = (bf16_t*)a.y2 + (ch - g.D1); dstride[h] = g.D2; }
        else { dst[h] = (bf16_t*)a.y1 + ch; dstride[h] = g.D2 ? g.D1 : CALL; }
    }
    float s1[STATS ? CL : 1], s2[STATS ? CL : 1];
    if (STATS) {
#pragma unroll
        for (int c = 0; c < CL; ++c) s1[c] = s2[c] = 0.f;
    }
    __syncthreads();
    const int64_t ngroups = ((int64_t)g.M + 15) / 16;
    const int64_t g0 = ((int64_t)blockIdx.x * (4 / NSPLIT) + pw) * groups_per_wave;
    const int64_t g1 = min(ngroups, g0 + groups_per_wave);

    auto load = [&](int64_t gb, bf8_t (*xb)[KB]) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int64_t pix = (gb + t) * 16 + fr;
            const int64_t pc = (gb + t < g1 && pix < g.M) ? pix : 0;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const bf16_t* src = kb * 32 < C1 ? x1 + pc * C1 + kb * 32 + fq * 8 : x2 + pc * C2 + (kb * 32 - C1) + fq * 8;
                xb[t][kb] = *(const bf8_t*)src;
            }
        }
    };
    auto compute = [&](int64_t gb, bf8_t (*xb)[KB]) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
            if (gb + t >= g1) break;
            const int64_t pix = (gb + t) * 16 + fr;
            const bool valid = pix < g.M;
            if (PRE) {
                // the producer's BatchNorm + ReLU on the fragment: the same fma -> max -> bf16 rounding as bn_apply_k
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) {
                    const uint4 u = __builtin_bit_cast(uint4, xb[t][kb]);
                    const unsigned wv[4] = {u.x, u.y, u.z, u.w};
                    const f4_t sc0 = *(const f4_t*)&spre[0][kb * 32 + fq * 8], sc1 = *(const f4_t*)&spre[0][kb * 32 + fq * 8 + 4];
                    const f4_t sh0 = *(const f4_t*)&spre[1][kb * 32 + fq * 8], sh1 = *(const f4_t*)&spre[1][kb * 32 + fq * 8 + 4];
                    const float sc[8] = {sc0[0], sc0[1], sc0[2], sc0[3], sc1[0], sc1[1], sc1[2], sc1[3]};
                    const float sh[8] = {sh0[0], sh0[1], sh0[2], sh0[3], sh1[0], sh1[1], sh1[2], sh1[3]};
                    unsigned o[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float lo = fmaxf(fmaf(__uint_as_float(wv[i] << 16), sc[2 * i], sh[2 * i]), plo);
                        const float hi = fmaxf(fmaf(__uint_as_float(wv[i] & 0xffff0000u), sc[2 * i + 1], sh[2 * i + 1]), plo);
                        o[i] = pk2bf(lo, hi);
                    }
                    xb[t][kb] = __builtin_bit_cast(bf8_t, make_uint4(o[0], o[1], o[2], o[3]));
                }
            }
            uint4 zq[MODE == 2 ? NCH : 1];
            if (MODE == 2) {        // the producer's raw output for this lane's chunks: in flight under the MFMAs
                const int64_t pc = valid ? pix : 0;
#pragma unroll
                for (int h = 0; h < NCH; ++h) zq[h] = *(const uint4*)(bzp + pc * CALL + cbase + 32 * h + 8 * fq);
            }
            f4_t acc[NTT];
#pragma unroll
            for (int nt = 0; nt < NTT; ++nt) {
                acc[nt] = *(const f4_t*)&sbias[cbase + 32 * (nt >> 1) + 8 * fq + 4 * (nt & 1)];
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt][kb], xb[t][kb], acc[nt], 0, 0, 0);
            }
            // lane: pixel `pix`, channels 32 (nt >> 1) + 8 fq + 4 (nt & 1) + r
#pragma unroll
            for (int h = 0; h < NCH; ++h) {
                unsigned pk[4];
                const unsigned zw[4] = {zq[MODE == 2 ? h : 0].x, zq[MODE == 2 ? h : 0].y, zq[MODE == 2 ? h : 0].z,
                                        zq[MODE == 2 ? h : 0].w};
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int nt = 2 * h + j;
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] = acc[nt][r];
                        if (MODE == 1 && valid) { s1[4 * nt + r] += v[r]; s2[4 * nt + r] = fmaf(v[r], v[r], s2[4 * nt + r]); }
                    }
                    if (MODE == 2) {
                        const int cb = cbase + 32 * h + 8 * fq + 4 * j;        // this tile's 4 channels
                        const f4_t bsc = *(const f4_t*)&sbwd[0][cb], bsh = *(const f4_t*)&sbwd[1][cb], bmu = *(const f4_t*)&sbwd[2][cb];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const unsigned w2 = zw[2 * j + (r >> 1)];
                            const float z = __uint_as_float((r & 1) ? (w2 & 0xffff0000u) : (w2 << 16));
                            if (brelu && !(fmaf(z, bsc[r], bsh[r]) > 0.f)) v[r] = 0.f;
                            const float dr = bf2f(f2bf(v[r]));          // sums of the value as stored (what pass 2 reads back)
                            if (valid) { s1[4 * nt + r] += dr; s2[4 * nt + r] = fmaf(dr, z - bmu[r], s2[4 * nt + r]); }
                        }
                    }
                    pk[2 * j] = pk2bf(v[0], v[1]);
                    pk[2 * j + 1] = pk2bf(v[2], v[3]);
                }
                // (eight groups per batch for the 64-channel inputs, i.e. twice the loads in flight: 350 against 339 us -- the launch is not
                //  bound by the latency of its reads; non-temporal loads / stores, which help the elementwise passes of ew_stream.hip, cost this kernel 5-15 %:
                //  396 against 346 us for 64 -> 128 channels at 512 x 512 x 16)
                if (valid) *(uint4*)(dst[h] + pix * dstride[h]) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            }
        }
    };

    bf8_t xa[T][KB], xc[T][KB];
    if (g0 < g1) load(g0, xa);
    for (int64_t gb = g0; gb < g1; gb += 2 * T) {
        if (gb + T < g1) load(gb + T, xc);
        compute(gb, xa);
        if (gb + 2 * T < g1) load(gb + 2 * T, xa);
        if (gb + T < g1) compute(gb + T, xc);
    }
    if (!STATS || (MODE == 2 && !a.bpart)) return;
    // sum over the 16 pixels of each lane row (DPP), then over the 4 waves; one partial row per workgroup
#pragma unroll
    for (int c = 0; c < CL; ++c) {
        const float u = dpp_add16(s1[c]), q = dpp_add16(s2[c]);
        const int ch = 32 * (c >> 3) + 8 * fq + (c & 7);       // c = 4 nt + r
        if (fr == 0) { sred[wid][0][ch] = u; sred[wid][1][ch] = q; }
    }
    __syncthreads();
    for (int c = tid; c < CALL; c += 256) {
        const int half = c / COUT, cc = c - half * COUT;       // the waves wid % NSPLIT == half hold channel c
        float u = 0.f, q = 0.f;
#pragma unroll
        for (int wv = 0; wv < 4 / NSPLIT; ++wv) { u += sred[wv * NSPLIT + half][0][cc]; q += sred[wv * NSPLIT + half][1][cc]; }
        float* row = (MODE == 2 ? a.bpart : a.stats) + (size_t)blockIdx.x * 2 * CALL;
        row[c] = u;
        row[CALL + c] = MODE == 2 ? a.brstd[c] * q : q;       // producer backward: sum du * xhat = rstd * sum du * (z - mean)
    }
}

template <int CIN, int COUT, int T, int NSPLIT = 1>
static void pwx_launch(const GG& g, const FwdArgs& a, int blocks, int gpw, hipStream_t s) {
    if (a.bz) {             // input gradient with the producer's backward in its store
        PAI_LAUNCH((pwx_k<CIN, COUT, T, 2, false, NSPLIT>), dim3(blocks), dim3(256), 0, s, g, a, gpw);
    } else if (a.pscale) {  // forward with the producer's BatchNorm on load
        if (a.stats) PAI_LAUNCH((pwx_k<CIN, COUT, T, 1, true, NSPLIT>), dim3(blocks), dim3(256), 0, s, g, a, gpw);
        else PAI_LAUNCH((pwx_k<CIN, COUT, T, 0, true, NSPLIT>), dim3(blocks), dim3(256), 0, s, g, a, gpw);
    } else {
        if (a.stats) PAI_LAUNCH((pwx_k<CIN, COUT, T, 1, false, NSPLIT>), dim3(blocks), dim3(256), 0, s, g, a, gpw);
        else PAI_LAUNCH((pwx_k<CIN, COUT, T, 0, false, NSPLIT>), dim3(blocks), dim3(256), 0, s, g, a, gpw);
    }
}
template <int CIN, int COUT>
static void pwx_launch_t(const GG& g, const FwdArgs& a, int blocks, int gpw, int t, hipStream_t s) {
    if (t >= 4) pwx_launch<CIN, COUT, 4>(g, a, blocks, gpw, s);
    else if (t >= 2) pwx_launch<CIN, COUT, 2>(g, a, blocks, gpw, s);
    else pwx_launch<CIN, COUT, 1>(g, a, blocks, gpw, s);
}

int launch_pwx(const GG& g, const FwdArgs& a, hipStream_t s) {
    const int blocks = pwx_rows(g);
    const int64_t ngroups = ((int64_t)g.M + 15) / 16;
    const int ci = g.Cin, co = g.Cout;
    const int nsplit = ci * co > 128 * 128 ? 2 : 1;          // pixel streams per workgroup: 4 / nsplit
    const int gpw = (int)((ngroups + (int64_t)blocks * (4 / nsplit) - 1) / ((int64_t)blocks * (4 / nsplit)));
    if (nsplit == 2) {
        if (ci == 128 && co == 256) pwx_launch<128, 128, 2, 2>(g, a, blocks, gpw, s);
        else if (ci == 256 && co == 128) pwx_launch<256, 64, 2, 2>(g, a, blocks, gpw, s);
        else {
            pai_set_error("launch_pwx: no instantiation for %d -> %d channels", ci, co);
            return 1;
        }
        PAI_LAUNCH_CHECK();
        return 0;
    }
    const int big = ci * co > 64 * 128;
    int t = pai_tunable("pwx_t", big ? 2 : 4);
    if (co == 256 && (a.stats || a.bz)) t = 1;                     // (the 64-register statistics on top of a 128-register filter)
    if (ci == 64 && co == 64) pwx_launch_t<64, 64>(g, a, blocks, gpw, t, s);
    else if (ci == 64 && co == 128) pwx_launch_t<64, 128>(g, a, blocks, gpw, t, s);
    else if (ci == 128 && co == 64) pwx_launch_t<128, 64>(g, a, blocks, gpw, t, s);
    else if (ci == 128 && co == 128) pwx_launch_t<128, 128>(g, a, blocks, gpw, t, s);
    else if (ci == 256 && co == 64) pwx_launch_t<256, 64>(g, a, blocks, gpw, t, s);
    else if (ci == 64 && co == 256) pwx_launch_t<64, 256>(g, a, blocks, gpw, t, s);
    else {
        pai_set_error("launch_pwx: no instantiation for %d -> %d channels", ci, co);
        return 1;
    }
    PAI_LAUNCH_CHECK();
    return 0;
}

const char* pwx_kernel_name(const GG& g) {
    static thread_local char buf[48];
    snprintf(buf, sizeof(buf), "pwx_k<%d, %d>", g.Cin, g.Cout);
    return buf;
}

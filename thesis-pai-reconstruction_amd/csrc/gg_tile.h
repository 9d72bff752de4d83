// Helpers shared by the matrix-core gather-GEMM translation units (gg_mfma.hip, gg_p2.hip): vector types, the
// LDS-DMA macro, tile-order / row-decoding helpers, the fused producer-backward store and the 2 x 2-window
// patch geometry.  Device symbols are per translation unit (-fno-gpu-rdc), hence the static zero line.
#pragma once
#include <stdlib.h>

#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf4_t;
typedef __attribute__((ext_vector_type(4))) float f4_t;
typedef __attribute__((ext_vector_type(2))) short s2_t;
typedef __attribute__((ext_vector_type(4))) unsigned u4_t;

constexpr int MBK = 64;   // K per iteration (one tap, 64 channels)

static __device__ uint4 g_zero_line[16];  // 256 B of zeros: source of padding / masked rows

#define GLDS16(gptr, lptr)                                                                      \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),    \
                                     (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

typedef __attribute__((ext_vector_type(8))) short s8_t;

#ifndef RELU_ABL
#define RELU_ABL 0     // timing ablation (results WRONG): 1 = no ReLU on the fragments
#endif
__device__ __forceinline__ bf8_t relu_frag(bf8_t f) {
    if (RELU_ABL) return f;
    // ReLU on packed bf16: as signed 16-bit integers every negative float is negative
    // (v_pk_max_i16 x4).  Done on one 8-lane vector: element-wise writes to a 4 x u32 vector in an
    // unrolled loop were folded to a broadcast of element 0 by hipcc 7.2.
    s8_t x = __builtin_bit_cast(s8_t, f);
    const s8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
    x = __builtin_elementwise_max(x, z);
    return __builtin_bit_cast(bf8_t, x);
}

__device__ __forceinline__ void decode_row(const GG& g, int m, int& n, int& gy, int& gx) {
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

// XCD-aware tile order: consecutive logical tiles (same A rows, neighbouring image rows) land on
// the same XCD / L2.  Bijective for any grid size (guide T1).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// Producer backward on one 16-B chunk (8 channels) of the staged bf16 gradient tile:
//   du = act1'(pre) * g + act2'(pre) * add,  pre = z * sc + sh per channel (affine) or z itself,
// rounded to bf16; with `sums` the BatchNorm-backward sums are accumulated from the value as stored
// (what pai_bn_bwd_apply reads back).  Same du as bn_bwd_reduce_k / act_bwd_k on the bf16-rounded gradient.
struct BwdParams { float sc[8], sh[8]; };
static __device__ const float g_unit_affine[16] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
__device__ __forceinline__ void bwd_load_params(const FwdArgs& a, int c, BwdParams& P) {   // c: a multiple of 8
    // unconditional vector loads (a producer without BatchNorm reads scale 1 / shift 0 from a constant line): a load
    // per element behind `a.bscale ? ... : 1.f` was sixteen scalar branches
    const float* sp = a.bscale ? a.bscale + c : g_unit_affine;
    const float* hp = a.bscale ? a.bshift + c : g_unit_affine + 8;
    const float4 s0 = *(const float4*)sp, s1 = *(const float4*)(sp + 4);
    const float4 h0 = *(const float4*)hp, h1 = *(const float4*)(hp + 4);
    P.sc[0] = s0.x; P.sc[1] = s0.y; P.sc[2] = s0.z; P.sc[3] = s0.w; P.sc[4] = s1.x; P.sc[5] = s1.y; P.sc[6] = s1.z; P.sc[7] = s1.w;
    P.sh[0] = h0.x; P.sh[1] = h0.y; P.sh[2] = h0.z; P.sh[3] = h0.w; P.sh[4] = h1.x; P.sh[5] = h1.y; P.sh[6] = h1.z; P.sh[7] = h1.w;
}
// (act_slope / act_fwd: branch-free activations, common.h)
__device__ __forceinline__ float bwd_sel(float g, bool pos, float slope) {   // act'(pre) * g
    const float sg = slope == 0.f ? 0.f : fmaf(g, slope, 0.f);      // ReLU: 0 also for an infinite gradient (not inf x 0)
    return pos ? g : sg;
}
// s2 accumulates du * z; the tile's sum of du * xhat is rstd * (s2 - mean * s1), formed once per channel in
// bwd_write_partials (3 vector-ALU operations per element fewer in a store that runs behind every MFMA loop).
// P.sc / P.sh are 1 / 0 when the producer has no BatchNorm (bwd_load_params), so pre = fma(z, sc, sh) always.
template <bool HAS_ADD, bool SUMS>
__device__ __forceinline__ uint4 bwd_chunk_t(uint4 gq, uint4 zq, uint4 aq, float sl1, float sl2, const BwdParams& P,
                                             float* s1, float* s2) {
    const unsigned gw[4] = {gq.x, gq.y, gq.z, gq.w}, zw[4] = {zq.x, zq.y, zq.z, zq.w}, aw[4] = {aq.x, aq.y, aq.z, aq.w};
    unsigned o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float g0 = __uint_as_float(gw[k] << 16), g1 = __uint_as_float(gw[k] & 0xffff0000u);
        const float z0 = __uint_as_float(zw[k] << 16), z1 = __uint_as_float(zw[k] & 0xffff0000u);
        const bool q0 = fmaf(z0, P.sc[2 * k], P.sh[2 * k]) > 0.f;
        const bool q1 = fmaf(z1, P.sc[2 * k + 1], P.sh[2 * k + 1]) > 0.f;
        float d0 = bwd_sel(g0, q0, sl1), d1 = bwd_sel(g1, q1, sl1);
        if (HAS_ADD) {
            d0 += bwd_sel(__uint_as_float(aw[k] << 16), q0, sl2);
            d1 += bwd_sel(__uint_as_float(aw[k] & 0xffff0000u), q1, sl2);
        }
        o[k] = pk2bf(d0, d1);
        if (SUMS) {
            const float r0 = __uint_as_float(o[k] << 16), r1 = __uint_as_float(o[k] & 0xffff0000u);
            s1[2 * k] += r0;
            s1[2 * k + 1] += r1;
            s2[2 * k] = fmaf(r0, z0, s2[2 * k]);
            s2[2 * k + 1] = fmaf(r1, z1, s2[2 * k + 1]);
        }
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}
// run-time form: one wave-uniform branch per CHUNK (not per element)
__device__ __forceinline__ uint4 bwd_chunk(uint4 gq, uint4 zq, uint4 aq, bool has_add, bool /*affine*/, bool sums,
                                           int act1, int act2, const BwdParams& P, float* s1, float* s2) {
    const float sl1 = act_slope(act1), sl2 = act_slope(act2);
    if (has_add) return sums ? bwd_chunk_t<true, true>(gq, zq, aq, sl1, sl2, P, s1, s2)
                             : bwd_chunk_t<true, false>(gq, zq, aq, sl1, sl2, P, s1, s2);
    return sums ? bwd_chunk_t<false, true>(gq, zq, aq, sl1, sl2, P, s1, s2)
                : bwd_chunk_t<false, false>(gq, zq, aq, sl1, sl2, P, s1, s2);
}
// Per-tile reduction of the chunk sums: lanes of a wave that own the same chunk (lane % CPR) first, then the
// waves through `sred` [NW][2][BN]; thread c < BN writes column c of the tile's partial row.
template <int BN, int CPR, int NW>
__device__ __forceinline__ void bwd_write_partials(float* sred, const float* s1, const float* s2, int tid,
                                                   float* row_dst, int D1, const float* mean, const float* rstd) {
    const int lane = tid & 63, wid = tid >> 6;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float a = s1[k], b = s2[k];
#pragma unroll
        for (int o = CPR; o < 64; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
        if (lane < CPR) {
            sred[(wid * 2 + 0) * BN + lane * 8 + k] = a;
            sred[(wid * 2 + 1) * BN + lane * 8 + k] = b;
        }
    }
    __syncthreads();
    if (tid < BN) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) { a += sred[(w * 2 + 0) * BN + tid]; b += sred[(w * 2 + 1) * BN + tid]; }
        row_dst[tid] = a;
        row_dst[D1 + tid] = rstd[tid] * (b - mean[tid] * a);   // sum du * xhat from sum du * z
    }
}

struct PatchGeo {
    int groups;                    // windows per phase: 1 or 4
    int TY, TX;                    // 8 x 16 tiles per image
    signed char by[4][4], bx[4][4];  // [phase][window] source offset of patch pixel (0,0) from (gy0*S, gx0*S)
    unsigned toff4[4][4];          // 4 x 8 bit: patch offset ty*17+tx of the window's taps
    unsigned wt4[4][4];            // 4 x 8 bit: weight tap slot of the window's taps
    // the same per phase, packed for the forward kernel's scalar registers (patch_geo_pack): weight tap slots 4 bits per
    // (window, tap) -- windows 0-1 in wt_lo, 2-3 in wt_hi -- and patch offsets (ty, tx) 2 bits per (window, tap)
    unsigned wt_lo[4], wt_hi[4], toff2[4];
};
constexpr int PATCH_W = 17;
// BM = 128: 8 x 16 output pixels, 4 waves;  BM = 256: 16 x 16 output pixels, 8 waves (4 x 2) -- the weight
// tile fill is then shared by twice the rows: 25 KB of fill and 128 KB of fragment reads per 2 x 512
// MFMA cycles, the first configuration whose LDS time (916 cycles) is below its matrix time (1024).
template <int BM, int WN = 2, int WPX = 64> struct PatchDims {   // WN: waves side by side along the channels; WPX: pixels per wave
    static constexpr int TH = BM / 16;
    static constexpr int PIX = (TH + 1) * PATCH_W;
    static constexpr int NTHR = BM / WPX * 64 * WN;          // threads of the workgroup
    static constexpr int RPP = NTHR / 8;                     // pixels per block-wide fill instruction
    static constexpr int PJ = (PIX + RPP - 1) / RPP;
    static constexpr int BYTES = PJ * RPP * 128;
};

static inline void patch_geo_pack(PatchGeo* pg) {
    for (int ph = 0; ph < 4; ++ph) {
        pg->wt_lo[ph] = pg->wt_hi[ph] = pg->toff2[ph] = 0;
        for (int q = 0; q < 4; ++q)
            for (int k = 0; k < 4; ++k) {
                const unsigned slot = (pg->wt4[ph][q] >> (8 * k)) & 15u, to = (pg->toff4[ph][q] >> (8 * k)) & 0xffu;
                if (q < 2) pg->wt_lo[ph] |= slot << (16 * q + 4 * k); else pg->wt_hi[ph] |= slot << (16 * (q - 2) + 4 * k);
                pg->toff2[ph] |= ((to / PATCH_W) * 2u + (to % PATCH_W)) << (8 * q + 2 * k);
            }
    }
}
static inline bool patch_geo_raw(const GG& g, int th, PatchGeo* pg, int tw);
static inline bool patch_geo(const GG& g, int th, PatchGeo* pg, int tw = 16) {   // th x tw: output pixels of a tile
    if (!patch_geo_raw(g, th, pg, tw)) return false;
    patch_geo_pack(pg);
    return true;
}
static inline bool patch_geo_raw(const GG& g, int th, PatchGeo* pg, int tw) {
    if ((g.OWg % tw) || (g.OHg % th)) return false;
    memset(pg, 0, sizeof(*pg));
    pg->TY = g.OHg / th;
    pg->TX = g.OWg / tw;
    if (g.S == 1 && g.ntaps == 4) {
        pg->groups = 1;
        for (int ph = 0; ph < g.nphase; ++ph) {
            int by = 127, bx = 127;
            for (int t = 0; t < 4; ++t) { by = g.dy[ph][t] < by ? g.dy[ph][t] : by; bx = g.dx[ph][t] < bx ? g.dx[ph][t] : bx; }
            unsigned seen = 0;
            for (int t = 0; t < 4; ++t) {
                const int ty = g.dy[ph][t] - by, tx = g.dx[ph][t] - bx;
                if (ty > 1 || tx > 1) return false;
                // canonical slot order: slot k of a window is its tap at patch offset (ty, tx) = (k >> 1, k & 1) -- the
                // forward kernel unrolls the four taps of a window with compile-time patch offsets
                const int k = ty * 2 + tx;
                if (seen & (1u << k)) return false;
                seen |= 1u << k;
                pg->toff4[ph][0] |= (unsigned)(ty * PATCH_W + tx) << (8 * k);
                pg->wt4[ph][0] |= (unsigned)g.wt[ph][t] << (8 * k);
            }
            if (seen != 15u) return false;
            pg->by[ph][0] = (signed char)by;
            pg->bx[ph][0] = (signed char)bx;
        }
        return true;
    }
    if (g.S == 2 && g.ntaps == 16 && g.nphase == 1) {
        pg->groups = 4;
        int ymin = 127, xmin = 127;
        for (int t = 0; t < 16; ++t) { ymin = g.dy[0][t] < ymin ? g.dy[0][t] : ymin; xmin = g.dx[0][t] < xmin ? g.dx[0][t] : xmin; }
        for (int q = 0; q < 4; ++q) {
            const int by = ymin + (q >> 1), bx = xmin + (q & 1);
            int n = 0;
            unsigned seen = 0;
            for (int t = 0; t < 16; ++t) {
                const int ry = g.dy[0][t] - by, rx = g.dx[0][t] - bx;
                if (ry < 0 || rx < 0 || (ry & 1) || (rx & 1)) continue;
                const int ty = ry / 2, tx = rx / 2;
                if (ty > 1 || tx > 1 || n == 4) return false;
                const int k = ty * 2 + tx;           // canonical slot order, see above
                if (seen & (1u << k)) return false;
                seen |= 1u << k;
                pg->toff4[0][q] |= (unsigned)(ty * PATCH_W + tx) << (8 * k);
                pg->wt4[0][q] |= (unsigned)g.wt[0][t] << (8 * k);
                ++n;
            }
            if (n != 4 || seen != 15u) return false;
            pg->by[0][q] = (signed char)by;
            pg->bx[0][q] = (signed char)bx;
        }
        return true;
    }
    return false;
}

// Kernel arguments: plain scalars only.  By-value structs with arrays (GG, PatchGeo) indexed by the run-time phase --
// and even `cond ? a.x2 : a.x1` on neighbouring fields -- made hipcc 7.2 keep every argument in scratch and re-load
// them inside the K loop (vector loads, each followed by the vmcnt(0) that drains the LDS-DMA pipeline).  The
// per-phase tables are therefore packed into 64-bit scalars and unpacked with shifts:
//   wby / wbx  4 bits per (phase, window): source offset of patch pixel (0,0) from (gy0*S, gx0*S), biased by 8
//   toff       2 bits (ty, tx) per (phase, window, tap)
//   wt[phase]  4 bits weight tap slot per (window, tap)
//   poy / pox  1 bit per phase
struct P2Prob {
    int H, W, C1, C2, Cin, Cout, S, nphase, OH, OW, OS, D1, D2, wtaps, relu1, relu2;
    int groups, TY, TX, mtiles, ntiles;
    unsigned long long wby, wbx, toff[2], wt[4];
    unsigned poy, pox;
};

static inline void p2_prob(const GG& g, const PatchGeo& pg, int mtiles, int ntiles, P2Prob* o) {
    memset(o, 0, sizeof(*o));
    o->H = g.H; o->W = g.W; o->C1 = g.C1; o->C2 = g.C2; o->Cin = g.Cin; o->Cout = g.Cout; o->S = g.S;
    o->nphase = g.nphase; o->OH = g.OH; o->OW = g.OW; o->OS = g.OS; o->D1 = g.D1; o->D2 = g.D2; o->wtaps = g.wtaps;
    o->relu1 = g.relu1; o->relu2 = g.relu2;
    o->groups = pg.groups; o->TY = pg.TY; o->TX = pg.TX; o->mtiles = mtiles; o->ntiles = ntiles;
    for (int ph = 0; ph < 4; ++ph) {
        for (int q = 0; q < 4; ++q) {
            const int e = ph * 4 + q;
            o->wby |= (unsigned long long)((pg.by[ph][q] + 8) & 15) << (4 * e);
            o->wbx |= (unsigned long long)((pg.bx[ph][q] + 8) & 15) << (4 * e);
            unsigned b8 = 0, b16 = 0;
            for (int t = 0; t < 4; ++t) {
                const unsigned toff = (pg.toff4[ph][q] >> (8 * t)) & 0xffu, wt = (pg.wt4[ph][q] >> (8 * t)) & 0xffu;
                b8 |= ((toff / PATCH_W) * 2 + (toff % PATCH_W)) << (2 * t);
                b16 |= (wt & 15u) << (4 * t);
            }
            o->toff[e >> 3] |= (unsigned long long)b8 << (8 * (e & 7));
            o->wt[ph] |= (unsigned long long)b16 << (16 * q);
        }
        o->poy |= (unsigned)(g.poy[ph] & 1) << ph;
        o->pox |= (unsigned)(g.pox[ph] & 1) << ph;
    }
}


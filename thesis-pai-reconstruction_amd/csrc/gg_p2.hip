// Pipelined patch-resident forward / input-gradient kernel for gfx950 ("p2"): one 512-thread workgroup per CU,
// 256 x 256 (16 x 16 output pixels x 256 channels) or 512 x 128 (32 x 16 pixels x 128 channels) output tile, eight
// waves of 128 pixels x 64 channels each, v_mfma_f32_16x16x32_bf16 with fp32 accumulators in VGPRs.
//
// Why another kernel next to gg_fwd_patch_k (gg_mfma.hip): that one hides L2 -> LDS latency by running two
// workgroups per CU, exposes a whole patch refill every fourth K-step and, with 64 x 64 wave tiles, asks the LDS for
// 0.9 cycles per matrix cycle (profiles/README.md).  Here
//   * the wave tile is 128 x 64: 12 fragment reads per 32 MFMAs instead of 16 (LDS read port 37 % busy at full
//     matrix rate instead of 50 %), and the weight tile is shared by twice the pixels (21 B of LDS-DMA fill per
//     matrix cycle and CU instead of 25);
//   * the source patch holds 32 channels (64 B per pixel) and is double-buffered, the weight tiles sit in a ring of
//     three: every LDS-DMA is issued two K-steps (~2 us) before its data is needed and retired with a COUNTED
//     s_waitcnt vmcnt + raw s_barrier (guide: "Pipelining across barriers"), one barrier per K-step of 64 MFMAs per
//     wave;
//   * fragment reads are software-pipelined through the barrier: the MFMAs of one tap run while the fragments of
//     the next tap are read, so no wave waits for the LDS behind a barrier.
// K-step = 2 taps of one 2 x 2 window x 32 channels: weight tile row = [tap a: 32 ch | tap b: 32 ch] = 128 B.
//
// STATUS (round 2, measured with scripts/micro/convbench on one MI355X, bit-exact against gg_fwd_patch_k on integer
// data): NOT the default.  decoders[4] forward, 137 GFLOP: gg_fwd_patch_k 122-127 us (1.08-1.12 PFLOP/s); this kernel
// 138 us with 8 waves / one workgroup per CU, 128 us with 4 waves / two per CU.  Compile-time ablations of the 8-wave
// variant (P2_ABL): MFMAs alone 107 us -- of which ~15 us are prologue + epilogue that nothing overlaps when every CU
// runs exactly one tile, i.e. the matrix pipe sustains ~1.5 PFLOP/s inside the loop (power-limited clock, not 2.5) --
// LDS-DMA + fragment reads alone 100 us, fragment reads + MFMA 109 us, LDS-DMA + MFMA 121 us: what the deeper
// pipeline buys is lost to (a) the exposed epilogue (33 MB of output leave the chip in one burst at the end),
// (b) LDS-DMA instructions in the MFMA stream (spreading them one per MFMA group made it slower still: 128 -> 133 us).
// gg_fwd_patch_k's four waves per SIMD hide both; it stays the default and this file documents the alternative.
// A fourth variant (MT = 4: 8 waves of 64 x 64, 128 VGPRs, two workgroups per CU = gg_fwd_patch_k's occupancy, but
// with this file's prefetched 32-channel patches) is 5-7 % slower than gg_fwd_patch_k too (decoders[4] forward 126 vs
// 119 us, input gradient 118 vs 110): the exposed patch refill of gg_fwd_patch_k is NOT what holds it at half the
// matrix peak.  Counters (scripts/micro/pmc_variant.sh): with LDS-DMA and fragment reads compiled out the loop still
// keeps the matrix pipe only 52-55 % busy at 2.35-2.39 GHz -- the SIMD's vector issue is the shared resource: an
// MFMA 16x16x32 holds it for 8 of its 16 cycles and every other vector instruction for 4 (guide, cycle constants),
// so more than two vector instructions per MFMA (ReLU-on-load alone is one) cap the pipe below its peak.
// Two more forms were measured and dropped: variant 5 (32 x 16 pixels x 128 channels, sixteen 64 x 64 waves, ONE
// 1024-thread workgroup per CU: 34 % fewer LDS-DMA bytes per FLOP) runs 10-15 % slower than gg_fwd_patch_k on
// decoders[5], D block 2 and encoders[2]; and the v_mfma_f32_32x32x16_bf16 form of variants 1, 2 and 4 (which a bare
// MFMA loop favours by 40 % once other instructions sit between the MFMAs, scripts/micro/mfma_issue.hip) ran 3-8 %
// slower than their 16x16x32 form -- that code was not kept.  Nor was a 16 x 16 pixels x 64 channels tile with four
// 64 x 64 waves for the 64-channel layers (decoders[6] forward 166 -> 196-207 us, input gradient of encoders[1]
// 88 -> 116, of D block 1 192 -> 224): gg_fwd_patch_k<128, 64>'s four to five small workgroups per CU win there.
//
// Serves the same reference call sites as gg_fwd_patch_k: the Conv2d k4 s2 p1 / ConvTranspose2d k4 s2 p1 layers of
// EncoderBlock / DecoderBlock (models/pix2pix.py:58-111), DiscriminatorBlock 1-3 (models/wrapper.py:229-232) and
// the input-gradient halves of their aten::convolution_backward calls.
#include <stddef.h>

#include "gg_tile.h"

constexpr int P2_CK = 32;   // channels per patch chunk

template <int TH, int NW, int MT> struct P2Dims {      // NW waves of MT pixel rows (16 pixels each) x 64 channels
    static constexpr int NTHR = NW * 64;
    static constexpr int WMW = TH / MT, WNW = NW / WMW;     // waves along pixels / channels
    static constexpr int BN = WNW * 64, BM = TH * 16;
    static constexpr int PIX = (TH + 1) * PATCH_W;
    static constexpr int PPI = NTHR / 4;                    // patch pixels (64 B) per block-wide fill instruction
    static constexpr int PJ = (PIX + PPI - 1) / PPI;
    // a wave's piece of a fill instruction is 16 pixels: pieces wholly behind the last patch pixel land in one shared
    // 1-KB dump area instead of padding both patch buffers to PJ * PPI pixels
    static constexpr int PIXR = (PIX + 15) / 16 * 16;
    static constexpr int PBYTES = PIXR * 64;
    static constexpr int RPI = NTHR / 8;                    // weight rows (128 B) per block-wide fill instruction
    static constexpr int BJ = BN / RPI;
    static constexpr int BBYTES = BN * 128;
    static constexpr size_t lds_bytes(int nring) {
        const size_t loop = 2 * (size_t)PBYTES + (size_t)nring * BBYTES + 1024;
        // staged tile + statistics rows: [WMW][2][BN] (forward) or [NW][2][BN] (bwd_write_partials)
        const size_t epi = (size_t)BM * (BN * 2 + 16) + (size_t)(WMW > NW ? WMW : NW) * 2 * BN * sizeof(float);
        return loop > epi ? loop : epi;
    }
};

#ifndef P2_DMA_SPREAD
#define P2_DMA_SPREAD 1     // 1: the step's LDS-DMA instructions in two batches (behind the barrier, two items later), 0: one
#endif
#ifndef P2_SETPRIO
#define P2_SETPRIO 0
#endif
#ifndef P2_ABL
#define P2_ABL 0            // timing ablations (results WRONG): 1 no LDS-DMA, 2 no MFMA, 4 no fragment reads
#endif

// TH: pixel rows of the tile (16 wide); NW: waves; MT: pixel rows per wave (8: 128 x 64 wave tiles in 256 VGPRs, two
// waves per SIMD; 4: 64 x 64 wave tiles in 128 VGPRs, four); NRING: weight tiles in the ring (3: every LDS-DMA two
// K-steps ahead, 2: one); RELU: some input tensor is read through ReLU (decoder blocks)
template <int TH, int NW, int MT, int NRING, bool RELU>
__global__ __launch_bounds__(NW * 64, MT == 4 ? 4 : 2) void gg_fwd_p2_k(P2Prob g, FwdArgs a) {
    const int mtiles = g.mtiles, ntiles = g.ntiles;
    struct { int groups, TY, TX; } pg = {g.groups, g.TY, g.TX};
    typedef P2Dims<TH, NW, MT> PD;
    constexpr int WMW = PD::WMW, WNW = PD::WNW, BN = PD::BN, BM = PD::BM, NTHR = PD::NTHR;
    constexpr int NT = 4, NI = 2 * MT;
    constexpr int PFD = MT == 4 ? 2 : 4;                  // pixel fragments in flight (register sets)
    constexpr int BAR = PFD == 4 ? NI - 3 : NI - 2;       // the step's barrier sits behind this item
    constexpr int PJ = PD::PJ, PIX = PD::PIX, PBYTES = PD::PBYTES, PPI = PD::PPI, PIXR = PD::PIXR;
    constexpr int BJ = PD::BJ, RPI = PD::RPI, BBYTES = PD::BBYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // [patch 0][patch 1][weights 0] .. [weights NRING-1][dump]
    constexpr int B_OFF = 2 * PBYTES, DUMP_OFF = B_OFF + NRING * BBYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WNW, wn = wid % WNW;
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int bn = bid % ntiles;
    bid /= ntiles;
    const int ph = bid % g.nphase;
    const int bm = bid / g.nphase;
    const int n0 = bn * BN;
    const int tpi = pg.TY * pg.TX;
    const int img = bm / tpi, trem = bm - img * tpi;
    const int gy0 = (trem / pg.TX) * TH, gx0 = (trem % pg.TX) * 16;

    const bf16_t* w = (const bf16_t*)a.w;
    const bf16_t* zero = (const bf16_t*)g_zero_line;
    const bf16_t* x1p = (const bf16_t*)a.x1;
    const bf16_t* x2p = (const bf16_t*)a.x2;
    const int gC1 = g.C1, gC2 = g.C2, grelu1 = g.relu1, grelu2 = g.relu2, gD1 = g.D1, gD2 = g.D2;
    bf16_t* y1p = (bf16_t*)a.y1;
    bf16_t* y2p = (bf16_t*)a.y2;
    // tables of this workgroup's phase: unpacked with scalar shifts where they are used (a select chain over four
    // precomputed values becomes a lookup table in scratch under hipcc 7.2)
    const unsigned wby16 = (unsigned)(g.wby >> (16 * ph)) & 0xffffu, wbx16 = (unsigned)(g.wbx >> (16 * ph)) & 0xffffu;
    const unsigned tpk32 = (unsigned)(((ph & 2) ? g.toff[1] : g.toff[0]) >> (32 * (ph & 1)));   // 8 bits per window
    const unsigned long long wpk = ph == 0 ? g.wt[0] : (ph == 1 ? g.wt[1] : (ph == 2 ? g.wt[2] : g.wt[3]));   // 16 bits per window
    const int poy = (int)((g.poy >> ph) & 1u), pox = (int)((g.pox >> ph) & 1u);
    auto win_by = [&](int q) __attribute__((always_inline)) -> int { return (int)((wby16 >> (4 * q)) & 15u) - 8; };
    auto win_bx = [&](int q) __attribute__((always_inline)) -> int { return (int)((wbx16 >> (4 * q)) & 15u) - 8; };
    // ---- patch fill map: thread -> (pixel p = 128 j + tid / 4, 16-B slot tid % 4); the two 32-B halves of a pixel
    // are swapped when bit 2 of its column px = p % 17 is set: every ds_read_b128 lane group (16 consecutive pixels of
    // one patch row, at either tap shift) then covers the 16 slots of a 256-B bank row once, and -- the swizzle not
    // depending on the row -- the eight pixel rows of a wave are one base address + immediates.
    const int psc = tid & 3, pl = tid >> 2;
    // per fill instruction, one register: bits 0-25 source pixel of the patch pixel at window offset (0, 0) (host:
    // N * H * W < 2^26), bits 26-27 this lane's 16-B chunk of the 32 channels, bits 28-31 inside the image for window q
    unsigned pixb[PJ];
#pragma unroll
    for (int j = 0; j < PJ; ++j) {
        const int p = j * PPI + pl;
        const int py = p / PATCH_W, px = p - py * PATCH_W;
        const int y = (gy0 + py) * g.S, x = (gx0 + px) * g.S;
        unsigned m = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int yy = y + win_by(q), xx = x + win_bx(q);
            if (q < pg.groups && p < PIX && (unsigned)yy < (unsigned)g.H && (unsigned)xx < (unsigned)g.W) m |= 1u << q;
        }
        pixb[j] = ((unsigned)((img * g.H + y) * g.W + x) & 0x3ffffffu) | ((unsigned)(psc ^ (((px >> 2) & 1) << 1)) << 26) | (m << 28);
    }
    // ---- weight fill map: thread -> (row 64 j + tid / 8, 16-B slot tid % 8), slot s of row r holds global chunk
    // s ^ ((r >> 1) & 7); chunks 0-3 = 32 channels of the step's first tap, 4-7 = of its second tap.  LDS row
    // rho = 16 nt + i of a wave's 64 channels holds channel 16 (i >> 2) + 4 nt + (i & 3): with the weights as the MFMA's
    // A operand a lane ends up with 16 CONSECUTIVE channels of one pixel (16-B pieces in the epilogue).
    const int bsr = tid >> 3;
    const int gchB = (tid & 7) ^ ((bsr >> 1) & 7);
    const int tapsel = gchB >> 2;
    // element offset of this lane's chunk from w (host: Cout * wtaps * Cin < 2^31)
    const unsigned wrow0 = (unsigned)(n0 + 16 * ((bsr & 15) >> 2) + 4 * (bsr >> 4) + (bsr & 3)) * (unsigned)(g.wtaps * g.Cin) + (unsigned)(gchB & 3) * 8u;
    const int wrow_stride = g.wtaps * g.Cin;          // elements per output channel
    // fill instruction j covers LDS rows j * RPI + bsr: channel offset of its first row from that of j = 0
    auto wrow_ch = [](int j) __attribute__((always_inline)) -> int { return ((j * RPI) >> 6) * 64 + (((j * RPI) & 63) >> 4) * 4; };

    // ---- fragment read addresses ---------------------------------------------------------------------------
    const int fr = lane & 15, fq = lane >> 4;
    const unsigned wbase = (unsigned)(B_OFF + (wn * 64 + fr) * 128);
    const unsigned wsw = (unsigned)(fr >> 1);

    f4_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f4_t){0.f, 0.f, 0.f, 0.f};

    const int cchunks = g.Cin / P2_CK;
    const int gsh = pg.groups == 4 ? 2 : 0;
    const int ngroups = cchunks << gsh;
    const int tot = 2 * ngroups;

    // one LDS-DMA instruction of the patch of group gi / of the weight tile of step s
    auto patch_piece = [&](int gi, int j) __attribute__((always_inline)) {
        const int c0 = (gi >> gsh) * P2_CK, q = gi & (pg.groups - 1);
        const bool second = c0 >= gC1;
        const bf16_t* src = second ? x2p : x1p;
        const int C = second ? gC2 : gC1;
        const int cofs = (second ? c0 - gC1 : c0) + (int)((pixb[j] >> 26) & 3u) * 8;
        const int dpix = win_by(q) * g.W + win_bx(q);
        if (P2_ABL & 1) return;
        const bf16_t* pv = src + ((size_t)(unsigned)((int)(pixb[j] & 0x3ffffffu) + dpix) * (unsigned)C + cofs);
        const bf16_t* pa = ((pixb[j] >> (28 + q)) & 1u) ? pv : zero;
        const int p0 = j * PPI + wid * 16;       // first pixel of this wave's piece (wave-uniform)
        GLDS16(pa, smem + (p0 < PIXR ? (gi & 1) * PBYTES + p0 * 64 : DUMP_OFF));
    };
    auto weight_piece = [&](int s, int slot, int j) __attribute__((always_inline)) {
        const int gi = s >> 1, h = s & 1;
        const int c0 = (gi >> gsh) * P2_CK, q = gi & (pg.groups - 1);
        const int wt = (int)((unsigned)(wpk >> (16 * q + 8 * h + 4 * tapsel)) & 15u);   // tap 2 h + tapsel of window q
        if (P2_ABL & 1) return;
        GLDS16(w + (size_t)(wrow0 + (unsigned)(wrow_ch(j) * wrow_stride + (wt * g.Cin + c0))), smem + B_OFF + slot * BBYTES + (j * RPI + wid * 8) * 128);
    };
    // fragment reads: weights of tap k (0 / 1) of the step in ring slot `slot`; pixels of patch row mt at base `pa`
    auto read_w = [&](int slot, int k, bf8_t (&wf)[NT]) {
        const unsigned ad = wbase + (unsigned)(slot * BBYTES) + ((((unsigned)(k * 4 + fq)) ^ wsw) << 4);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (P2_ABL & 4) wf[nt] = __builtin_bit_cast(bf8_t, make_uint4(ad, nt, slot, k));
            else wf[nt] = *(const bf8_t*)(smem + ad + nt * 2048);
        }
    };
    // base address (patch row wm * MT, this lane's pixel and k-quarter) for tap offset toff = ty * 17 + tx of buffer b
    auto patch_base = [&](int b, int toff) __attribute__((always_inline)) -> unsigned {
        const int ty = toff >= PATCH_W ? 1 : 0;
        const unsigned px = (unsigned)(toff - ty * PATCH_W + fr);
        return (unsigned)(b * PBYTES) + (((unsigned)((wm * MT + ty) * PATCH_W) + px) << 6) + ((((unsigned)fq) ^ (((px >> 2) & 1u) << 1)) << 4);
    };
    auto read_p = [&](unsigned base, int mt) __attribute__((always_inline)) -> bf8_t {
        if (P2_ABL & 4) return __builtin_bit_cast(bf8_t, make_uint4(base, mt, base, mt));
        return *(const bf8_t*)(smem + base + mt * (PATCH_W * 64));
    };
    auto tap_off = [&](int s, int k) __attribute__((always_inline)) -> int {
        const int gi = s >> 1;
        const unsigned b2 = (tpk32 >> (8 * (gi & (pg.groups - 1)) + 2 * (2 * (s & 1) + k))) & 3u;   // (ty, tx) of the tap
        return (int)((b2 >> 1) * PATCH_W + (b2 & 1u));
    };
    // ReLU on load as a packed signed-16-bit max against a per-step threshold: 0 (ReLU) or -32768 (identity)
    auto relu_of = [&](int s) __attribute__((always_inline)) -> int { return (((s >> 1) >> gsh) * P2_CK >= gC1 ? grelu2 : grelu1) ? 0 : (int)0x80008000u; };
    auto mma = [&](const bf8_t (&wf)[NT], bf8_t x, int relu, int mt) {
        if (RELU) {
            typedef __attribute__((ext_vector_type(4))) int i4_t;
            i4_t xi = __builtin_bit_cast(i4_t, x);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int r;
                asm("v_pk_max_i16 %0, %1, %2" : "=v"(r) : "v"(xi[e]), "s"(relu));
                xi[e] = r;
            }
            x = __builtin_bit_cast(bf8_t, xi);
        }
        if (P2_ABL & 2) { acc[mt][0][0] += (float)x[0] + (float)wf[0][0] + (float)wf[1][1] + (float)wf[2][2] + (float)wf[3][3]; return; }
        if (P2_SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], x, acc[mt][nt], 0, 0, 0);
        if (P2_SETPRIO) __builtin_amdgcn_s_setprio(0);
    };

    // ---- prologue: P[0], W[0] | W[1] .. W[NRING-1], P[1] in flight ------------------------------------------------
#pragma unroll
    for (int j = 0; j < PJ; ++j) patch_piece(0, j);
#pragma unroll
    for (int r = 0; r < NRING; ++r) {
#pragma unroll
        for (int j = 0; j < BJ; ++j) weight_piece(r, r, j);
    }
#pragma unroll
    for (int j = 0; j < PJ; ++j) patch_piece(1, j);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NRING - 1) * BJ + PJ) : "memory");
    __builtin_amdgcn_s_barrier();

    // The K loop is a stream of items (step, tap, pixel row mt): NI = 2 MT per step, 4 MFMAs each.  Pixel fragments
    // live in a ring of four register sets: the fragment of item i + 4 is read right after the MFMAs of item i.  The
    // step's barrier sits behind item BAR = NI - 3: by then every read of W[s] and (odd steps) P[gi] has been issued
    // (the last one behind item NI - 5) -- lgkmcnt(0) retires them -- and the data of step s + 1 is retired by the
    // counted vmcnt.  Fragments of the next step that items NI - 4 .. BAR would have fetched are read right behind the
    // barrier instead.
    bf8_t wfA[NT], wfB[NT], pf[PFD];
    read_w(0, 0, wfA);
    unsigned pb_cur = patch_base(0, tap_off(0, 0));    // tap 0 of the current step
#pragma unroll
    for (int i = 0; i < PFD; ++i) pf[i] = read_p(pb_cur, i);
    int slot = 0;                                       // ring slot of W[s]
    for (int s = 0; s < tot; ++s) {
        const int gi = s >> 1;
        const int relu = relu_of(s);
        const bool more = s + 1 < tot;
        const int gn = (s + 1) >> 1;                       // group of the next step
        const int slot_n = slot == NRING - 1 ? 0 : slot + 1;       // W[s + 1]
        const unsigned pb_t1 = patch_base(gi & 1, tap_off(s, 1));
        const unsigned pb_next = more ? patch_base(gn & 1, tap_off(s + 1, 0)) : 0u;
        const bool wmore = s + NRING < tot;               // W[s+NRING] -> the slot of W[s]
        const bool pmore = (s & 1) && gi + 2 < ngroups;   // P[gi+2] -> the buffer of P[gi]
        // LDS-DMA pieces of W[s+NRING] / P[gi+2]
#define P2_DMA_PIECE(k)                                                          \
    do {                                                                         \
        if ((k) < BJ) { if (wmore) weight_piece(s + NRING, slot, (k)); }         \
        else if ((k) - BJ < PJ) { if (pmore) patch_piece(gi + 2, (k) - BJ); }    \
    } while (0)
        read_w(slot, 1, wfB);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int mt = i % MT;
            if (i < MT) mma(wfA, pf[i % PFD], relu, mt); else mma(wfB, pf[i % PFD], relu, mt);
            const int nx = i + PFD;                     // the item whose fragment goes into the register set just freed
            if (nx < NI) {
                pf[i % PFD] = read_p(nx < MT ? pb_cur : pb_t1, nx % MT);
            } else if (i == BAR) {
                if (more) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    // newer than what step s + 1 needs: the weight tiles W[s+2] .. W[s+NRING-1] and, behind an odd
                    // step's barrier, a patch (each batch is issued weights first)
                    if (s + 2 >= tot) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    else if (s & 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NRING - 2) * BJ) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NRING - 2) * BJ + PJ) : "memory");
                    __builtin_amdgcn_s_barrier();
                    read_w(slot_n, 0, wfA);
#pragma unroll
                    for (int d = NI - PFD; d <= BAR; ++d) pf[d % PFD] = read_p(pb_next, d + PFD - NI);
                    if (P2_DMA_SPREAD) {
#pragma unroll
                        for (int k = 0; k < (BJ + PJ + 1) / 2; ++k) P2_DMA_PIECE(k);
                    } else {
#pragma unroll
                        for (int k = 0; k < BJ + PJ; ++k) P2_DMA_PIECE(k);
                    }
                }
            } else if (i > BAR && more) {   // the last items of the step fetch for the next one
                pf[i % PFD] = read_p(pb_next, nx - NI);
                if (P2_DMA_SPREAD && i == BAR + 1) {
#pragma unroll
                    for (int k = (BJ + PJ + 1) / 2; k < BJ + PJ; ++k) P2_DMA_PIECE(k);
                }
            }
        }
        pb_cur = pb_next;
        slot = slot_n;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // the epilogue reuses the tile memory

    // ---- epilogue: bias, BN partial statistics, activation, LDS-staged row stores (as gg_fwd_patch_k) ----------
    constexpr int CROW = BN * 2 + 16;
    unsigned char* Cs = smem;
    float* sstat = (float*)(smem + BM * CROW);  // [WMW][2][BN]
    const int eact = a.yact ? a.eact : PAI_ACT_NONE;
    // lane (fq, fr) holds, for each of its MT pixel rows mt (pixel fr of the row), the 16 consecutive channels
    // wn*64 + 16 fq + (4 nt + r)
    constexpr int CL = 4 * NT;
    const int col0 = wn * 64 + CL * fq;
    float csum[CL], csq[CL];
    {
        float bias_v[CL];
#pragma unroll
        for (int c = 0; c < CL; ++c) { bias_v[c] = a.bias ? a.bias[n0 + col0 + c] : 0.f; csum[c] = csq[c] = 0.f; }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int row = wm * (MT * 16) + mt * 16 + fr;
            unsigned pk[CL / 2];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = acc[mt][nt][r] + bias_v[4 * nt + r];
                    csum[4 * nt + r] += v[r];
                    csq[4 * nt + r] = fmaf(v[r], v[r], csq[4 * nt + r]);
                    if (eact == PAI_ACT_LRELU) v[r] = fmaxf(v[r], 0.2f * v[r]);
                    else if (eact == PAI_ACT_RELU) v[r] = fmaxf(v[r], 0.f);
                }
                pk[2 * nt] = pk2bf(v[0], v[1]);
                pk[2 * nt + 1] = pk2bf(v[2], v[3]);
            }
#pragma unroll
            for (int h = 0; h < CL / 8; ++h)
                *(uint4*)(Cs + row * CROW + (col0 + 8 * h) * 2) = make_uint4(pk[4 * h], pk[4 * h + 1], pk[4 * h + 2], pk[4 * h + 3]);
        }
    }
    if (a.stats) {
        // sum over the 16 pixels (lanes fr) of every row of 16 lanes: quad_perm, row_half_mirror, row_mirror
#pragma unroll
        for (int c = 0; c < CL; ++c) {
            float s = csum[c], q = csq[c];
            s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0xB1, 0xF, 0xF, false));
            q += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, q), 0xB1, 0xF, 0xF, false));
            s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x4E, 0xF, 0xF, false));
            q += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, q), 0x4E, 0xF, 0xF, false));
            s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x141, 0xF, 0xF, false));
            q += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, q), 0x141, 0xF, 0xF, false));
            s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x140, 0xF, 0xF, false));
            q += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, q), 0x140, 0xF, 0xF, false));
            if (fr == 0) {
                sstat[(wm * 2 + 0) * BN + col0 + c] = s;
                sstat[(wm * 2 + 1) * BN + col0 + c] = q;
            }
        }
    }
    __syncthreads();
    if (a.stats && tid < BN) {
        float* dst = a.stats + ((size_t)(ph * mtiles + bm) * 2) * g.Cout + n0 + tid;
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int i = 0; i < WMW; ++i) { s += sstat[(i * 2 + 0) * BN + tid]; q += sstat[(i * 2 + 1) * BN + tid]; }
        dst[0] = s;
        dst[g.Cout] = q;
    }
    bf16_t* dst;
    int dstride, dcol;
    if (a.yact) { dst = (bf16_t*)a.yact; dstride = g.Cout; dcol = n0; }
    else if (n0 < gD1) { dst = y1p; dstride = gD1; dcol = n0; }
    else { dst = y2p; dstride = gD2; dcol = n0 - gD1; }
    const bool bwd = a.bz && !a.yact && n0 < gD1;   // uniform per workgroup
    const bf16_t* bzp = (const bf16_t*)a.bz;
    const bf16_t* bap = (const bf16_t*)a.badd;
    const bool bsum = bwd && a.bpart;
    constexpr int CPR = BN / 8;        // 16-B chunks per row
    constexpr int ORP = NTHR / CPR;    // rows per pass
    const int oc = tid % CPR, orow0 = tid / CPR;
    BwdParams BP;
    float bs1[8], bs2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) bs1[k] = bs2[k] = 0.f;
    // producer chunks requested four passes at a time, ahead of that batch's stores (see gg_fwd_mfma_k)
    constexpr int NP = BM / ORP, NB = 4;
    if (bwd) bwd_load_params(a, dcol + oc * 8, BP);
#pragma unroll 1
    for (int p0 = 0; p0 < NP; p0 += NB) {
        size_t offs[NB];
        uint4 zq[NB], aq[NB];
#pragma unroll
        for (int p = 0; p < NB; ++p) {
            const int row = orow0 + (p0 + p) * ORP;
            const int gy = gy0 + (row >> 4), gx = gx0 + (row & 15);
            const size_t pix = (size_t)(img * g.OH + gy * g.OS + poy) * g.OW + gx * g.OS + pox;
            offs[p] = pix * dstride + dcol + oc * 8;
            if (bwd) {
                zq[p] = *(const uint4*)(bzp + offs[p]);
                aq[p] = bap ? *(const uint4*)(bap + offs[p]) : make_uint4(0, 0, 0, 0);
            }
        }
#pragma unroll
        for (int p = 0; p < NB; ++p) {
            const int row = orow0 + (p0 + p) * ORP;
            uint4 o = *(const uint4*)(Cs + row * CROW + oc * 16);
            if (bwd)
                o = bwd_chunk(o, zq[p], aq[p], bap != nullptr, a.bscale != nullptr, bsum, a.bact1, a.bact2, BP, bs1, bs2);
            *(uint4*)(dst + offs[p]) = o;
        }
    }
    if (bsum)
        bwd_write_partials<BN, CPR, NTHR / 64>(sstat, bs1, bs2, tid,
                                               a.bpart + ((size_t)(ph * mtiles + bm) * 2) * g.D1 + n0, g.D1,
                                               a.bmean + n0, a.brstd + n0);
}

// ---- host side --------------------------------------------------------------------------------------------
// Tile choice for a problem the matrix-core path accepts (fwd_mfma_ok).  Returns 0 (not for this kernel) or the
// variant: 1 = 16 x 16 pixels x 256 channels, 8 waves; 2 = 32 x 16 pixels x 128 channels, 8 waves (one workgroup per
// CU each); 3 = 16 x 16 pixels x 128 channels, 4 waves, two workgroups per CU; 4 = the same tile with 8 waves of
// 64 x 64 (128 VGPRs), two workgroups per CU = four waves per SIMD like gg_fwd_patch_k.
// tunable "fwd_p2": 0 (default) off, 1 the 8-wave variants, 2 the 4-wave variant only, 3 the 4-wave variant where it
// applies, else the 8-wave ones, 4 the 64 x 64-wave variant only.  OFF by default: measured on MI355X (scripts/micro/convbench, round 2) every variant is
// 3-25 % SLOWER than gg_fwd_patch_k on the layers it accepts -- see the header of this file and DESIGN.md.
static int p2_variant(const GG& g) {
    const int mode = pai_tunable("fwd_p2", 0);
    if (!mode) return 0;
    if ((g.C1 % P2_CK) || (g.C2 % P2_CK) || g.Cin < 64) return 0;
    if (((int64_t)g.N * g.H + 4) * (g.W + 4) >= (1 << 26) || (int64_t)g.Cout * g.wtaps * g.Cin >= (1ll << 31)) return 0;   // packed offsets
    PatchGeo pg;
    const bool c128 = (g.Cout % 128) == 0 && (g.D2 == 0 || (g.D1 % 128) == 0);
    const bool c256 = (g.Cout % 256) == 0 && (g.D2 == 0 || (g.D1 % 256) == 0);
    // 5: 32 x 16 pixels x 128 channels, 16 waves of 64 x 64 (128 VGPRs), ONE workgroup per CU: the weight tile and the
    // patch halo are shared by twice the pixels -- 4.1 B of LDS-DMA fill per kFLOP instead of gg_fwd_patch_k's 6.2
    if (mode == 5) {
        if (c128 && patch_geo(g, 32, &pg) && (int64_t)(g.M / 512) * (g.Cout / 128) * g.nphase >= pai_tunable("fwd_p2_min_wgs5", 200)) return 5;
        return 0;
    }
    if (mode >= 2 && c128 && patch_geo(g, 16, &pg) &&
        (int64_t)(g.M / 256) * (g.Cout / 128) * g.nphase >= pai_tunable("fwd_p2_min_wgs4", 384))
        return mode == 4 ? 4 : 3;
    if (mode == 2 || mode == 4) return 0;
    const int min_wgs = pai_tunable("fwd_p2_min_wgs", 200);
    if (c256 && patch_geo(g, 16, &pg) && (int64_t)(g.M / 256) * (g.Cout / 256) * g.nphase >= min_wgs) return 1;
    if (c128 && patch_geo(g, 32, &pg) && (int64_t)(g.M / 512) * (g.Cout / 128) * g.nphase >= min_wgs) return 2;
    return 0;
}

int fwd_p2_rows(const GG& g) {
    const int v = p2_variant(g);
    return v == 0 ? 0 : ((v == 2 || v == 5) ? 512 : 256);
}

template <int TH, int NW, int MT, int NRING>
static int p2_launch(const GG& g, const FwdArgs& a, const PatchGeo& pg, hipStream_t s) {
    typedef P2Dims<TH, NW, MT> PD;
    const size_t lds = PD::lds_bytes(NRING);
    static bool attr = false;
    if (!attr) {
        const void* fns[2] = {reinterpret_cast<const void*>(&gg_fwd_p2_k<TH, NW, MT, NRING, false>),
                              reinterpret_cast<const void*>(&gg_fwd_p2_k<TH, NW, MT, NRING, true>)};
        for (int i = 0; i < 2; ++i) {
            hipError_t e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            PAI_CHECK(e == hipSuccess, "hipFuncSetAttribute(max dynamic LDS): %s", hipGetErrorString(e));
        }
        attr = true;
    }
    const int mtiles = g.M / PD::BM, ntiles = g.Cout / PD::BN;
    P2Prob pr;
    p2_prob(g, pg, mtiles, ntiles, &pr);
    const dim3 grid(mtiles * ntiles * g.nphase), block(PD::NTHR);
    if (g.relu1 || g.relu2) PAI_LAUNCH((gg_fwd_p2_k<TH, NW, MT, NRING, true>), grid, block, lds, s, pr, a);
    else PAI_LAUNCH((gg_fwd_p2_k<TH, NW, MT, NRING, false>), grid, block, lds, s, pr, a);
    PAI_LAUNCH_CHECK();
    return 0;
}

int launch_fwd_p2(const GG& g, const FwdArgs& a, hipStream_t s) {
    const int v = p2_variant(g);
    PatchGeo pg;
    PAI_CHECK(v && patch_geo(g, (v == 2 || v == 5) ? 32 : 16, &pg), "launch_fwd_p2: problem not eligible");
    if (v == 5) return pai_tunable("fwd_p2_ring5", 3) == 3 ? p2_launch<32, 16, 4, 3>(g, a, pg, s) : p2_launch<32, 16, 4, 2>(g, a, pg, s);
    if (v == 1) return p2_launch<16, 8, 8, 3>(g, a, pg, s);
    if (v == 2) return p2_launch<32, 8, 8, 3>(g, a, pg, s);
    if (v == 4) return p2_launch<16, 8, 4, 2>(g, a, pg, s);
    return p2_launch<16, 4, 8, 2>(g, a, pg, s);
}

const char* fwd_p2_kernel_name(const GG& g) {
    const bool relu = g.relu1 || g.relu2;
    switch (p2_variant(g)) {
        case 1: return relu ? "gg_fwd_p2_k<16, 8, 8, 3, true>" : "gg_fwd_p2_k<16, 8, 8, 3, false>";
        case 2: return relu ? "gg_fwd_p2_k<32, 8, 8, 3, true>" : "gg_fwd_p2_k<32, 8, 8, 3, false>";
        case 5: return pai_tunable("fwd_p2_ring5", 3) == 3 ? (relu ? "gg_fwd_p2_k<32, 16, 4, 3, true>" : "gg_fwd_p2_k<32, 16, 4, 3, false>")
                                                           : (relu ? "gg_fwd_p2_k<32, 16, 4, 2, true>" : "gg_fwd_p2_k<32, 16, 4, 2, false>");
        case 4: return relu ? "gg_fwd_p2_k<16, 8, 4, 2, true>" : "gg_fwd_p2_k<16, 8, 4, 2, false>";
        default: return relu ? "gg_fwd_p2_k<16, 4, 8, 2, true>" : "gg_fwd_p2_k<16, 4, 8, 2, false>";
    }
}

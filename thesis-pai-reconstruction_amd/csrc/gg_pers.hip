// Persistent patch-resident forward / input-gradient kernel for gfx950 (round 5): gg_fwd_pers_k.
//
// Same matrix loop as gg_fwd_patch_k (gg_mfma.hip): a workgroup owns a 16 x 16 (8 x 16) block of output pixels, keeps
// the source pixels of a 2 x 2 tap window in LDS (one patch fill serves four taps), streams 64-channel weight tiles
// through a two-slot ring, v_mfma_f32_16x16x32_bf16 with the weights as the A operand so that a lane ends up with
// 4 NT CONSECUTIVE output channels of one pixel.  What is new is everything around that loop:
//   * a workgroup walks SEVERAL tiles (tile = blockIdx.x + k * gridDim.x in the XCD-aware order): index setup, buffer
//     descriptors and fragment addresses are paid once per workgroup, not once per tile;
//   * the epilogue never touches the LDS: bias, BatchNorm partial statistics, activation and the fused backward of the
//     producing layer (pai_conv_dgrad_act / _bn: z and the second gradient are LOADED for exactly the 16 channels x
//     4 pixels a lane holds) run on the accumulator registers and leave through 16-B global stores -- no staging
//     pass, no barrier, partial statistics one row per WAVE ROW (tile x 4) straight to memory;
//   * so the LDS is free the moment the last fragment read of a tile has retired, and the first patch and weight
//     tile of the NEXT tile are in flight (LDS-DMA) while the epilogue of this one computes and stores; the next
//     tile's first K step waits for the fills only (counted s_waitcnt: the stores behind them stay in flight).
// Measured against gg_fwd_patch_k in scripts/micro/convbench (--set fwd_pers=0 --set fwd_pers=1, bit-exact on integer
// data) and in the step (bench.py --set fwd_pers=...); numbers in DESIGN.md section 12.
//
// Serves the same reference call sites as gg_fwd_patch_k: the Conv2d k4 s2 p1 / ConvTranspose2d k4 s2 p1 layers of
// EncoderBlock / DecoderBlock (models/pix2pix.py:58-111), DiscriminatorBlock 1-3 (models/wrapper.py:229-232) and the
// input-gradient halves of their aten::convolution_backward calls.
#include <stdlib.h>

#include <type_traits>

#include "gg_tile.h"

#ifndef PERS_ABL
#define PERS_ABL 0   // timing ablations (results WRONG): 16 no epilogue
#endif

// sum over the 16 lanes fr of every row of 16 lanes (quad_perm, quad_perm, row_half_mirror, row_mirror): every lane
// of the row ends up with the row's total
__device__ __forceinline__ float row16_sum(float s) {
    s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0xB1, 0xF, 0xF, false));
    s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x4E, 0xF, 0xF, false));
    s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x141, 0xF, 0xF, false));
    s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x140, 0xF, 0xF, false));
    return s;
}

// "this value exists in registers HERE": empty asm statements are not reordered against each other, so a chain of pins
// fixes the order in which the epilogue's loads, roundings and stores happen (left alone, hipcc keeps the accumulators
// alive beside every load buffer of the unrolled epilogue and spills ~400 registers)
__device__ __forceinline__ void pin(uint4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }

// bwd_chunk (gg_tile.h) with the activation kinds as slopes instead of switches: act'(pre) * g = pos ? g : neg(g),
// neg(g) = slope * g, and exactly +0 for ReLU (slope * g would be -0 for a negative g).  Same values bit for bit.
struct BwdAct { float slope; bool zero; };
__device__ __forceinline__ BwdAct bwd_act(int act) {
    BwdAct r;
    r.slope = act == PAI_ACT_LRELU ? 0.2f : 1.f;
    r.zero = act == PAI_ACT_RELU;
    return r;
}
__device__ __forceinline__ float bwd_sel2(float g, bool pos, const BwdAct& A) {
    const float n = A.zero ? 0.f : A.slope * g;
    return pos ? g : n;
}
__device__ __forceinline__ uint4 bwd_chunk_bf(uint4 gq, uint4 zq, uint4 aq, bool has_add, bool affine, bool sums,
                                              const BwdAct& A1, const BwdAct& A2, const BwdParams& P, float* s1, float* s2) {
    const unsigned gw[4] = {gq.x, gq.y, gq.z, gq.w}, zw[4] = {zq.x, zq.y, zq.z, zq.w}, aw[4] = {aq.x, aq.y, aq.z, aq.w};
    unsigned o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float g0 = __uint_as_float(gw[k] << 16), g1 = __uint_as_float(gw[k] & 0xffff0000u);
        const float z0 = __uint_as_float(zw[k] << 16), z1 = __uint_as_float(zw[k] & 0xffff0000u);
        // (affine == false: scale 1, shift 0 -- fmaf(z, 1, 0) == z exactly)
        const bool q0 = fmaf(z0, P.sc[2 * k], P.sh[2 * k]) > 0.f;
        const bool q1 = fmaf(z1, P.sc[2 * k + 1], P.sh[2 * k + 1]) > 0.f;
        float d0 = bwd_sel2(g0, q0, A1), d1 = bwd_sel2(g1, q1, A1);
        // (has_add == false: the caller passes zeros, and x + (+-0) == x for every x the first term can be ... except
        //  x = -0: -0 + +0 = +0.  So the sum is only formed when there IS a second gradient, as in bwd_chunk.)
        const float e0 = d0 + bwd_sel2(__uint_as_float(aw[k] << 16), q0, A2);
        const float e1 = d1 + bwd_sel2(__uint_as_float(aw[k] & 0xffff0000u), q1, A2);
        d0 = has_add ? e0 : d0;
        d1 = has_add ? e1 : d1;
        o[k] = pk2bf(d0, d1);
        const float r0 = __uint_as_float(o[k] << 16), r1 = __uint_as_float(o[k] & 0xffff0000u);
        s1[2 * k] += r0;
        s1[2 * k + 1] += r1;
        s2[2 * k] = fmaf(r0, z0, s2[2 * k]);
        s2[2 * k + 1] = fmaf(r1, z1, s2[2 * k + 1]);
    }
    (void)affine; (void)sums;
    return make_uint4(o[0], o[1], o[2], o[3]);
}

// BM: output pixels of a tile (256: 16 x 16, eight waves; 128: 8 x 16, four waves); BN: output channels of a tile
// (two wave columns of BN / 2); DBB: two weight-tile buffers
// All kernel arguments in ONE by-value struct = the kernel-argument segment.  The matrix loop reads its fields through
// the parameter (hipcc hoists those loads out of the tile loop, as it should); the per-tile setup and the epilogue read
// theirs through the segment pointer, laundered once per tile (KARGS): hoisted as well, the ~60 scalars they need
// would be live across the matrix loop -- 92 spilled SGPRs and 780 spilled VGPRs in the first build of this kernel.
struct PersArgs {
    GG g;
    FwdArgs a;
    PatchGeo pg;
    int mrows, ntiles, total;
    int stagger;   // the upper half of the grid starts this many ~2 us naps late (see launch_pers_e)
};
typedef const __attribute__((address_space(4))) PersArgs* PersArgsPtr;
#define KARGS(name)                                                        \
    PersArgsPtr name = (PersArgsPtr)__builtin_amdgcn_kernarg_segment_ptr(); \
    asm volatile("" : "+s"(name))

// Per-tile state of a thread: everything the fills of a tile need.  Rebuilt from the (laundered) thread index
// whenever it is needed -- twice per tile: once in front of the previous tile's epilogue, to put this tile's first patch
// and weight tile in flight, and again behind that epilogue for the matrix loop.  Kept alive ACROSS the epilogue these
// registers (and the thread-position constants they derive from) were spilled by hipcc and reloaded inside the matrix
// loop, each reload followed by the s_waitcnt vmcnt(0) that drains the LDS-DMA pipeline.
template <int PJ, int BJ> struct PersTile {
    int ph, bm, n0, img, gy0, gx0;
    unsigned wby16, wbx16;
    unsigned pfill[PJ];
    unsigned wrow[BJ];
};

// EPI: 0 forward epilogue (bias, BatchNorm partial statistics, activation), 1 fused backward of the producing layer in
// the store of the D1 part (pai_conv_dgrad_act / _bn), plain store of the D2 part
template <int BM, int BN, bool DBB, int EPI>
__device__ __forceinline__ void gg_fwd_pers_body(const PersArgs& P) {
    const GG& g = P.g;
    const FwdArgs& a = P.a;
    const PatchGeo& pg = P.pg;
    const int total = P.total;
    constexpr int abl = PERS_ABL;
    typedef PatchDims<BM, 2, 64> PD;
    constexpr int MT = 4, NT = BN / 32;
    constexpr int BNW = BN / 2;              // channels per wave column
    constexpr int WM = BM / 64;              // wave rows
    constexpr int RPP = PD::RPP, PJ = PD::PJ, PATCH_PIX = PD::PIX, PATCH_BYTES = PD::BYTES;
    constexpr int BJ = BN / RPP;             // weight tile fill instructions per thread
    static_assert(BJ >= 1, "weight tile smaller than one block-wide fill instruction");
    typedef PersTile<PJ, BJ> Tile;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Bs = smem + PATCH_BYTES;

    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wid >> 1, wn = wid & 1;

    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(a.w), 0, (unsigned)(g.Cout * g.wtaps * g.Cin) * 2u, 0x00020000);
    const unsigned xpix = (unsigned)(g.N * g.H * g.W);
    const __amdgpu_buffer_rsrc_t x1rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(a.x1), 0, xpix * (unsigned)g.C1 * 2u, 0x00020000);
    const __amdgpu_buffer_rsrc_t x2rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(a.x2 ? a.x2 : a.x1), 0, a.x2 ? xpix * (unsigned)g.C2 * 2u : 0u, 0x00020000);

    const int cchunks = g.Cin / MBK;
    const int ngroups = cchunks * pg.groups;
    const int gsh = pg.groups == 4 ? 2 : 0;

    // patch fill: thread -> (pixel p = RPP j + tid / 8, 16-B slot tid % 8); slot c of patch pixel p holds chunk
    // c ^ (p & 6) (scripts/lds_swizzle_check.py fwd_patch)
    auto setup = [&](int vb, Tile& T) {
        KARGS(K);
        int t_ = threadIdx.x;
        asm volatile("" : "+v"(t_));   // (opaque: nothing below may be hoisted out of the tile loop and kept alive)
        const int sc = t_ & 7, sr = t_ >> 3;
        const int nphase = K->g.nphase, S = K->g.S, H = K->g.H, W = K->g.W;
        const int tpi = K->pg.TY * K->pg.TX, TX = K->pg.TX, groups = K->pg.groups;
        int bid = xcd_remap(vb, K->total);
        const int bn = bid % K->ntiles;
        bid /= K->ntiles;
        T.ph = bid % nphase;
        T.bm = bid / nphase;
        T.n0 = bn * BN;
        T.img = T.bm / tpi;
        const int trem = T.bm - T.img * tpi;
        T.gy0 = (trem / TX) * PD::TH;
        T.gx0 = (trem % TX) * 16;
        // the four window offsets of this phase: one dword of by[4][4] / bx[4][4] each (scalar loads)
        typedef const __attribute__((address_space(4))) unsigned* U4;
        const unsigned byw = ((U4)&K->pg.by[0][0])[T.ph], bxw = ((U4)&K->pg.bx[0][0])[T.ph];
        unsigned by16 = 0, bx16 = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            by16 |= (((byw >> (8 * q)) + 8u) & 15u) << (4 * q);
            bx16 |= (((bxw >> (8 * q)) + 8u) & 15u) << (4 * q);
        }
        T.wby16 = __builtin_amdgcn_readfirstlane(by16);
        T.wbx16 = __builtin_amdgcn_readfirstlane(bx16);
        // per fill instruction ONE register: bits 0-23 source pixel index of the patch pixel for window offset (0, 0),
        // bits 24-27 "inside the image" for window q of this phase, bits 28-30 this thread's chunk
#pragma unroll
        for (int j = 0; j < PJ; ++j) {
            const int p = j * RPP + sr;
            const int py = p / PATCH_W, px = p - py * PATCH_W;
            const int y = (T.gy0 + py) * S, x = (T.gx0 + px) * S;
            const int pixb = (T.img * H + y) * W + x;
            unsigned m = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int yy = y + (int)((T.wby16 >> (4 * q)) & 15u) - 8, xx = x + (int)((T.wbx16 >> (4 * q)) & 15u) - 8;
                if (q < groups && p < PATCH_PIX && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) m |= 1u << q;
            }
            T.pfill[j] = ((unsigned)pixb & 0xffffffu) | (m << 24) | ((unsigned)(sc ^ (sr & 6)) << 28);
        }
        // LDS row rho = 16 nt + i of a wave's half of the weight tile holds output channel
        // 32 (nt >> 1) + 8 (i >> 2) + 4 (nt & 1) + (i & 3) of that half: with the weights as the MFMA's A operand lane
        // (fq, fr) ends up with the 8 consecutive channels 8 fq .. 8 fq + 7 of every 32-channel group h = nt >> 1 of pixel
        // fr, i.e. the four lanes fq of a pixel hold 64 contiguous bytes per group: the 16-B stores (and the fused
        // backward's loads) of one instruction cover whole 64-B half lines.  (gg_fwd_patch_k gives a lane 16 consecutive
        // channels, which its LDS staging wants; stored straight from the registers that layout leaves 16-B pieces 32 B
        // apart: +8-12 % per launch, first build of this kernel.)
        const int wtc = K->g.wtaps * K->g.Cin;
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int lr = sr + RPP * j, half = lr / BNW, rho = lr % BNW;
            const int ch = half * BNW + 32 * (rho >> 5) + 8 * ((rho & 15) >> 2) + 4 * ((rho >> 4) & 1) + (rho & 3);
            T.wrow[j] = (unsigned)((T.n0 + ch) * wtc + (sc ^ ((sr >> 1) & 7)) * 8) * 2u;
        }
    };

#define PS_BLDS16(rs, voff, soff, lptr) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lptr), 16, (int)(voff), (int)(soff), 0, 0)
    auto fire_patch = [&](const Tile& T, int gi) {
        const int c0 = (gi >> gsh) * MBK, q = gi & (pg.groups - 1);
        const bool second = c0 >= g.C1;
        const int C = second ? g.C2 : g.C1;
        const int cofs = second ? c0 - g.C1 : c0;
        const int dpix = ((int)((T.wby16 >> (4 * q)) & 15u) - 8) * g.W + (int)((T.wbx16 >> (4 * q)) & 15u) - 8;
#pragma unroll
        for (int j = 0; j < PJ; ++j) {
            unsigned pf = T.pfill[j];
            asm volatile("" : "+v"(pf));
            const unsigned vo = ((pf >> (24 + q)) & 1u)
                                    ? (unsigned)(((int)(pf & 0xffffffu) + dpix) * C + cofs + (int)((pf >> 28) & 7u) * 8) * 2u : OOB;
            if (second) PS_BLDS16(x2rs, vo, 0, smem + (j * RPP + wid * 8) * 128);
            else PS_BLDS16(x1rs, vo, 0, smem + (j * RPP + wid * 8) * 128);
        }
    };
    auto fire_b = [&](const Tile& T, int gi, int k, int buf) {
        const int c0 = (gi >> gsh) * MBK, q = gi & (pg.groups - 1);
        const unsigned woff = (unsigned)((int)((pg.wt4[T.ph][q] >> (8 * k)) & 0xffu) * g.Cin + c0) * 2u;
#pragma unroll
        for (int j = 0; j < BJ; ++j) PS_BLDS16(wrs, T.wrow[j], woff, Bs + buf * (BN * 128) + (j * RPP + wid * 8) * 128);
    };

    // every wave issues at least this many vector-memory operations behind the next tile's first fills: the 16-B
    // stores of its accumulator tile
    constexpr int CL = 4 * NT;               // channels per lane
    constexpr int NST = MT * (CL / 8);

    int vb = blockIdx.x;
    bool pend = false;
    // Workgroups of one launch start together and take the same time per tile, so all of them reach their epilogue --
    // the memory-bound part of a tile -- at the same moment: the chip alternates between a matrix phase and an HBM
    // phase instead of overlapping the two.  The second workgroup of every CU (the upper half of the grid, as the
    // dispatcher deals them) therefore starts half a tile late and stays out of phase with its neighbour.
    if (P.stagger > 0 && blockIdx.x >= (gridDim.x >> 1)) {
        for (int i = 0; i < P.stagger; ++i) __builtin_amdgcn_s_sleep(64);
    }
    {
        Tile T0;
        setup(vb, T0);
        fire_patch(T0, 0);
        fire_b(T0, 0, 0, 0);
    }
    for (;;) {
        Tile T;
        setup(vb, T);
        // fragment read addresses (from the laundered thread index, see PersTile)
        int t_ = threadIdx.x;
        asm volatile("" : "+v"(t_));
        const int fr = t_ & 15, fq = (t_ >> 4) & 3;
        const int fswz = fr >> 1;
        int pbase[MT];                           // patch pixel of the wave's row mt (tap offset added per step)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) pbase[mt] = (wm * MT + mt) * PATCH_W + fr;
        const unsigned b_base = (unsigned)(PATCH_BYTES + (wn * BNW + fr) * 128);

        f4_t acc[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = (f4_t){0.f, 0.f, 0.f, 0.f};

        int buf = 0;
        for (int gi = 0; gi < ngroups; ++gi) {
            const int c0g = (gi >> gsh) * MBK;
            const int relu = c0g >= g.C1 ? g.relu2 : g.relu1;
            const unsigned toff4 = pg.toff4[T.ph][gi & (pg.groups - 1)];
            const bool more = gi + 1 < ngroups;
#pragma unroll 1
            for (int k = 0; k < 4; ++k) {
                const int toff = (int)((toff4 >> (8 * k)) & 0xffu);
                unsigned abase[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const unsigned pp = (unsigned)(pbase[mt] + toff);
                    abase[mt] = (pp << 7) ^ ((pp & 6u) << 4);
                }
                // this step's tiles have landed.  First step of a later tile: only the fills count -- the stores of the
                // previous tile's epilogue were issued behind them and stay in flight.
                if (pend && gi == 0 && k == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();   // ... for every wave; everyone is done with the other weight buffer
                if (DBB) {
                    if (k < 3) fire_b(T, gi, k + 1, buf ^ 1);
                    else if (more) fire_b(T, gi + 1, 0, buf ^ 1);
                }
                const unsigned bb = b_base + (DBB ? buf * (BN * 128) : 0);
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const unsigned ca = (unsigned)((kk * 4 + fq) << 4);
                    const unsigned cb = (unsigned)(((kk * 4 + fq) ^ fswz) << 4);
                    bf8_t af[MT], bfr[NT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) af[mt] = *(const bf8_t*)(smem + (abase[mt] ^ ca));
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) bfr[nt] = *(const bf8_t*)(smem + bb + nt * 16 * 128 + cb);
                    if (relu) {
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) af[mt] = relu_frag(af[mt]);
                    }
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            // D[i = channel slot][j = pixel]: acc[mt][nt][r] = channel slot 4 fq + r of pixel fr
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[nt], af[mt], acc[mt][nt], 0, 0, 0);
                }
                if (DBB) {
                    buf ^= 1;
                    if (k == 3 && more) {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();   // every wave is done reading the patch
                        fire_patch(T, gi + 1);
                    }
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();       // every wave is done reading before the next fill overwrites
                    if (k < 3) fire_b(T, gi, k + 1, 0);
                    else if (more) { fire_patch(T, gi + 1); fire_b(T, gi + 1, 0, 0); }
                }
            }
        }
        if (DBB) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();   // every wave is done reading: the LDS may take the next tile's first fills
        }

        // ---- this tile's output coordinates (scalars), then the next tile's first fills ------------
        const int e_ph = T.ph, e_bm = T.bm, e_n0 = T.n0, e_img = T.img, e_gy0 = T.gy0, e_gx0 = T.gx0;
        const int nvb = vb + (int)gridDim.x;
        const bool next = nvb < total;
        if (next) {
            Tile Tn;
            setup(nvb, Tn);
            fire_patch(Tn, 0);
            fire_b(Tn, 0, 0, 0);
        }

        if (abl & 16) {   // every accumulator stays live, nothing of the epilogue runs
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
            if (t == 123.456f) *(float*)a.y1 = t;
        } else {
            // ---- epilogue on the accumulator registers -----------------------------------------------
            // lane (fq, fr) holds, for each of its 4 pixel rows mt and each 32-channel group h, the 8 channels
            // col0 + 32 h .. + 7 of pixel (gy0 + 4 wm + mt, gx0 + fr); processed one group (a 16-B piece per pixel) at a time
            KARGS(E);
            int u_ = threadIdx.x;
            asm volatile("" : "+v"(u_));   // (opaque copy, as in setup)
            const int er = u_ & 15, eq = (u_ >> 4) & 3;
            void* const yact = E->a.yact;
            const int D1 = E->g.D1, OS = E->g.OS, OW = E->g.OW;
            const int col0 = wn * BNW + 8 * eq;
            bf16_t* dst;
            int dstride, dcol;
            if (yact) { dst = (bf16_t*)yact; dstride = E->g.Cout; dcol = e_n0; }
            else if (e_n0 < D1) { dst = (bf16_t*)E->a.y1; dstride = D1; dcol = e_n0; }
            else { dst = (bf16_t*)E->a.y2; dstride = E->g.D2; dcol = e_n0 - D1; }
            const int gy = e_gy0 + wm * MT, gx = e_gx0 + er;
            const size_t pix0 = (size_t)(e_img * E->g.OH + gy * OS + E->g.poy[e_ph]) * OW + gx * OS + E->g.pox[e_ph];
            const size_t off0 = pix0 * dstride + dcol + col0;
            const size_t rstep = (size_t)OS * OW * dstride;      // one tile row down
            const size_t prow = (size_t)(e_ph * E->mrows + e_bm * WM + wm);   // partial-statistics row of this wave row
            const bool bwd = EPI == 1 && e_n0 < D1;   // uniform per workgroup
            if (!bwd) {
                const float* bias = EPI == 0 ? E->a.bias : nullptr;
                float* stats = EPI == 0 ? E->a.stats : nullptr;
                const int eact = (EPI == 0 && yact) ? E->a.eact : PAI_ACT_NONE;
                // (the activation is a compile-time tag: as a run-time value inside the unrolled loops it became a scalar
                //  branch per element, with ~800 register moves around them)
                auto fwd_epi = [&](auto act_tag) {
                    constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
                for (int h = 0; h < CL / 8; ++h) {
                    float bias_v[8], csum[8], csq[8];
#pragma unroll
                    for (int c = 0; c < 8; ++c) { bias_v[c] = bias ? bias[e_n0 + col0 + 32 * h + c] : 0.f; csum[c] = csq[c] = 0.f; }
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        unsigned pk[4];
#pragma unroll
                        for (int n2 = 0; n2 < 2; ++n2) {
                            float v[4];
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                v[r] = acc[mt][2 * h + n2][r] + bias_v[4 * n2 + r];
                                if (EPI == 0) {
                                    csum[4 * n2 + r] += v[r];
                                    csq[4 * n2 + r] = fmaf(v[r], v[r], csq[4 * n2 + r]);
                                    if (ACT == PAI_ACT_LRELU) v[r] = fmaxf(v[r], 0.2f * v[r]);
                                    else if (ACT == PAI_ACT_RELU) v[r] = fmaxf(v[r], 0.f);
                                }
                            }
                            pk[2 * n2] = pk2bf(v[0], v[1]);
                            pk[2 * n2 + 1] = pk2bf(v[2], v[3]);
                        }
                        *(uint4*)(dst + off0 + mt * rstep + 32 * h) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                    }
                    if (EPI == 0 && stats) {
                        const int Cout = E->g.Cout;
                        float* srow = stats + prow * 2 * Cout + e_n0 + col0 + 32 * h;
#pragma unroll
                        for (int c = 0; c < 8; ++c) { csum[c] = row16_sum(csum[c]); csq[c] = row16_sum(csq[c]); }
                        if (er == 0) {
                            *(float4*)(srow) = make_float4(csum[0], csum[1], csum[2], csum[3]);
                            *(float4*)(srow + 4) = make_float4(csum[4], csum[5], csum[6], csum[7]);
                            *(float4*)(srow + Cout) = make_float4(csq[0], csq[1], csq[2], csq[3]);
                            *(float4*)(srow + Cout + 4) = make_float4(csq[4], csq[5], csq[6], csq[7]);
                        }
                    }
                }
                };
                if (eact == PAI_ACT_LRELU) fwd_epi(std::integral_constant<int, PAI_ACT_LRELU>{});
                else if (eact == PAI_ACT_RELU) fwd_epi(std::integral_constant<int, PAI_ACT_RELU>{});
                else fwd_epi(std::integral_constant<int, PAI_ACT_NONE>{});
            } else {
                // fused backward of the producing layer: same values as gg_fwd_patch_k's staged form (the product is
                // formed from the bf16-rounded gradient).  Per 8-channel group: the z / second-gradient pieces of all MT
                // pixel rows are requested first, the accumulators of the group rounded meanwhile.
                const bf16_t* bzp = (const bf16_t*)E->a.bz;
                const bf16_t* bap = (const bf16_t*)E->a.badd;
                const float* bscale = E->a.bscale;
                const float* bshift = E->a.bshift;
                float* bpart = E->a.bpart;
                const BwdAct A1 = bwd_act(E->a.bact1), A2 = bwd_act(E->a.bact2);
                const bool has_add = bap != nullptr, affine = bscale != nullptr, bsum = bpart != nullptr;
                // order of work (the register budget is what is left beside 64 accumulators at four waves per SIMD):
                // request the pieces of channel group 0 for all MT pixel rows; round ALL accumulators to bf16 meanwhile
                // (64 -> 32 registers); then, row by row, finish a piece of group h and request the same row's piece of
                // group h + 1 into the registers that just became free.
                constexpr int NH = CL / 8;
                uint4 zq[MT], aq[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    zq[mt] = *(const uint4*)(bzp + off0 + mt * rstep);
                    aq[mt] = has_add ? *(const uint4*)(bap + off0 + mt * rstep) : make_uint4(0, 0, 0, 0);
                }
                uint4 gq[NH][MT];
#pragma unroll
                for (int h = 0; h < NH; ++h)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const f4_t v0 = acc[mt][2 * h], v1 = acc[mt][2 * h + 1];
                        gq[h][mt] = make_uint4(pk2bf(v0[0], v0[1]), pk2bf(v0[2], v0[3]), pk2bf(v1[0], v1[1]), pk2bf(v1[2], v1[3]));
                        pin(gq[h][mt]);
                    }
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    BwdParams BP;
                    const int c = dcol + col0 + 32 * h;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        BP.sc[k] = affine ? bscale[c + k] : 1.f;
                        BP.sh[k] = affine ? bshift[c + k] : 0.f;
                    }
                    float bs1[8], bs2[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) bs1[k] = bs2[k] = 0.f;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        uint4 o = bwd_chunk_bf(gq[h][mt], zq[mt], aq[mt], has_add, affine, bsum, A1, A2, BP, bs1, bs2);
                        pin(o);
                        *(uint4*)(dst + off0 + mt * rstep + 32 * h) = o;
                        if (h + 1 < NH) {
                            zq[mt] = *(const uint4*)(bzp + off0 + mt * rstep + 32 * (h + 1));
                            aq[mt] = has_add ? *(const uint4*)(bap + off0 + mt * rstep + 32 * (h + 1)) : make_uint4(0, 0, 0, 0);
                        }
                    }
                    if (bsum) {
                        float* prow_dst = bpart + prow * 2 * D1 + c;
                        const float* bmean = E->a.bmean;
                        const float* brstd = E->a.brstd;
#pragma unroll
                        for (int k = 0; k < 8; ++k) { bs1[k] = row16_sum(bs1[k]); bs2[k] = row16_sum(bs2[k]); }
                        if (er == 0) {
                            float t2[8];
#pragma unroll
                            for (int k = 0; k < 8; ++k) t2[k] = brstd[c + k] * (bs2[k] - bmean[c + k] * bs1[k]);   // sum du * xhat from sum du * z
                            *(float4*)(prow_dst) = make_float4(bs1[0], bs1[1], bs1[2], bs1[3]);
                            *(float4*)(prow_dst + 4) = make_float4(bs1[4], bs1[5], bs1[6], bs1[7]);
                            *(float4*)(prow_dst + D1) = make_float4(t2[0], t2[1], t2[2], t2[3]);
                            *(float4*)(prow_dst + D1 + 4) = make_float4(t2[4], t2[5], t2[6], t2[7]);
                        }
                    }
                }
            }
        }
        if (!next) break;
        vb = nvb;
        // the fused-backward form waits for its own loads (issued behind the fills), statistics rows may add stores on
        // some lanes: in every form at least NST stores follow the fills, see the first step's wait
        pend = true;
    }
}

template <int BM, int BN, bool DBB, int EPI>
__global__ __launch_bounds__(BM * 2, BM == 256 ? 4 : (BN == 64 ? 4 : 3)) void gg_fwd_pers_k(PersArgs P) {
    gg_fwd_pers_body<BM, BN, DBB, EPI>(P);
}

// ---- host side -------------------------------------------------------------------------------------
// bm: 256 (16 x 16 tiles, eight waves) or 128 (8 x 16, four waves); bn: 128 or 64.  The caller (launch_fwd_mfma) has
// checked the patch geometry and the 2 GB limits of the buffer descriptors.
bool fwd_pers_ok(int bm, int bn) {
    // bit 0: 256 x 128 tiles, bit 1: 128 x 64 tiles, bit 2: 128 x 128 tiles
    const int mode = pai_tunable("fwd_pers", 0);
    if (bm == 256 && bn == 128) return (mode & 1) != 0;
    if (bm == 128 && bn == 64) return (mode & 2) != 0;
    if (bm == 128 && bn == 128) return (mode & 4) != 0;
    return false;
}

// BatchNorm partial-statistics rows per phase: one per wave row (64 output pixels)
int fwd_pers_rows(const GG& g) { return g.M / 64; }

// (the symbol carries the epilogue form as a fourth template argument: 0 forward, 1 fused producer backward)
const char* fwd_pers_kernel_name(int bm, int bn, bool db) {
    if (bm == 256) return db ? "gg_fwd_pers_k<256, 128, true" : "gg_fwd_pers_k<256, 128, false";
    if (bn == 128) return db ? "gg_fwd_pers_k<128, 128, true" : "gg_fwd_pers_k<128, 128, false";
    return db ? "gg_fwd_pers_k<128, 64, true" : "gg_fwd_pers_k<128, 64, false";
}

template <int BM, int BN, bool DBB, int EPI>
static int launch_pers_e(const GG& g, const FwdArgs& a, const PatchGeo& pg, int wgs_per_cu, hipStream_t s) {
    typedef PatchDims<BM, 2, 64> PD;
    const size_t lds = PD::BYTES + (size_t)BN * 128 * (DBB ? 2 : 1);
    static PerDeviceOnce attr;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (lds > 64 * 1024 && attr.first()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gg_fwd_pers_k<BM, BN, DBB, EPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        PAI_CHECK(e == hipSuccess, "hipFuncSetAttribute(max dynamic LDS): %s", hipGetErrorString(e));
    }
    const int mtiles = g.M / BM, ntiles = g.Cout / BN;
    const int total = mtiles * ntiles * g.nphase;
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    int grid = cus * wgs_per_cu;
    if (grid > total) grid = total;
    PersArgs P;
    P.g = g; P.a = a; P.pg = pg;
    P.mrows = fwd_pers_rows(g); P.ntiles = ntiles; P.total = total;
    // half a tile's matrix loop: a K step of 64 takes ~0.45 us per workgroup when two share a CU
    const int ksteps = g.ntaps * g.Cin / MBK;
    const int stag = pai_tunable("pers_stagger", 0);   // percent of a tile's matrix loop; 0: off
    P.stagger = (grid >= 2 * cus / 2 && grid > cus) ? (int)(ksteps * 0.45 * stag / 100.0 / 1.7) : 0;
    PAI_LAUNCH((gg_fwd_pers_k<BM, BN, DBB, EPI>), dim3(grid), dim3(BM * 2), lds, s, P);
    PAI_LAUNCH_CHECK();
    return 0;
}

template <int BM, int BN, bool DBB>
static int launch_pers_t(const GG& g, const FwdArgs& a, const PatchGeo& pg, int wgs_per_cu, hipStream_t s) {
    // fused backward of the producer in the store (pai_conv_dgrad_act / _bn) or the forward epilogue
    if (a.bz && !a.yact) return launch_pers_e<BM, BN, DBB, 1>(g, a, pg, wgs_per_cu, s);
    return launch_pers_e<BM, BN, DBB, 0>(g, a, pg, wgs_per_cu, s);
}

int launch_fwd_pers(const GG& g, const FwdArgs& a, const PatchGeo& pg, int bm, int bn, bool db, hipStream_t s) {
    // workgroups per CU of the grid: every workgroup walks total / grid tiles
    if (bm == 256) {
        const int wpc = pai_tunable("pers_wpc", 2);
        return db ? launch_pers_t<256, 128, true>(g, a, pg, wpc, s) : launch_pers_t<256, 128, false>(g, a, pg, wpc, s);
    }
    if (bn == 128) {
        const int wpc = pai_tunable("pers_wpc128", 3);
        return db ? launch_pers_t<128, 128, true>(g, a, pg, wpc, s) : launch_pers_t<128, 128, false>(g, a, pg, wpc, s);
    }
    const int wpc = pai_tunable("pers_wpc64", 5);
    return db ? launch_pers_t<128, 64, true>(g, a, pg, wpc, s) : launch_pers_t<128, 64, false>(g, a, pg, wpc, s);
}

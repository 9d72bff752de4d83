// Patch-resident forward / input-gradient kernel whose WEIGHTS never touch the LDS ("bd": B operand direct).
// 256-thread workgroup, 16 x 16 output pixels x 128 channels, four waves of 128 pixels x 64 channels
// (v_mfma_f32_16x16x32_bf16, 128 fp32 accumulator registers per lane), two workgroups per CU.
//
// gg_fwd_patch_k (gg_mfma.hip) is bound by the LDS: per K-step every wave re-reads both operands from it, the weight
// tile is written into it by LDS-DMA, and a workgroup barrier per K-step keeps its eight waves in lock-step
// (128 us with everything, 106 without the weight-tile fill, 77 with neither fills nor fragment
// reads, decoders[4]).  Here
//   * the source patch (32 channels, 64 B per pixel, double-buffered) is the only thing in the LDS: one fill of
//     18.5 KB per 4 taps and ONE barrier per 4 taps (128 MFMAs per wave);
//   * weights come from a FRAGMENT-MAJOR copy of the filter pack (pai_pack_frag): for every 64-channel tile and
//     every 32-deep K slice the four 16 x 32 MFMA operand fragments of a wave are 4 KB of contiguous memory in lane
//     order, so a wave fetches its operand with four fully coalesced buffer_load_dwordx4 straight into registers,
//     one K slice ahead of the MFMAs that use it (two register sets) -- no LDS write, no LDS read, no barrier;
//   * a wave's eight pixel-row fragments are one LDS base address + immediates (the swizzle depends on the patch
//     column only): 8 ds_read_b128 + 4 buffer loads per 32 MFMAs.
// The two waves of a workgroup that share a channel half fetch the same weights (L1 / L2 hits): 25-31 B/clk/CU of
// vector-memory return traffic at full matrix rate, under the 64 B/clk of the path.  A second form (WIDE, tunable
// fwd_bd = 2) puts the four waves side by side along the channels (256 pixels x 32 channels each): no weight byte is
// fetched twice, every wave reads the whole patch.
//
// STATUS (round 2, scripts/micro/convbench --frag, one MI355X, bit-exact against gg_fwd_patch_k on integer data and in
// tests/test_gpu_conv_exact.py): OPT-IN (needs the fragment-major pack copy, pai_conv_desc.pack_flags) and NOT faster:
//   us, gg_fwd_patch_k / this kernel 2 x 2 / WIDE     forward            input gradient
//   decoders[4]  (137 GFLOP)                          117.1 122.8 122.9  108.5 110.9 110.7
//   D block 3                                         113.0 115.0 114.2  111.5 115.6 113.9
//   decoders[5]                                       122.7 130.5 132.8  113.4 119.7 115.7
// Three very different operand paths end within 5 % of each other.  What was learnt on the way:
//   * ablations (input gradient of decoders[4], a slower box): everything 125 us, no weight loads 106, no patch fill
//     113, no fragment reads 120, MFMAs + epilogue alone 92: with the weights out of the LDS the fragment reads are
//     cheap (5 us) and the cost moves to the vector-memory path;
//   * counters (scripts/micro/pmc_variant.sh): this kernel keeps the matrix pipe 46 % busy at a 14 % HIGHER clock than
//     gg_fwd_patch_k (54 % busy): the chip trades clock for activity, the product is the same 1.1-1.15;
//   * scripts/micro/mfma_mix.hip: the instruction mix of a tap (32 MFMAs + 8 ds_read_b128 + 4 buffer loads, two waves
//     per SIMD, everything L1-resident) sustains 1.43 PFLOP/s with the reads behind consecutive MFMAs, 1.55-1.56 with
//     reads and loads clustered, 1.64 with one of them behind every third MFMA (bare MFMAs 1.86; the 32x32x16 form
//     1.43-1.50 / 1.69): the mix is not what holds the kernel at 1.2;
//   * hipcc: builtin MFMAs / loads in a hand-pipelined loop of 128 accumulator registers get renamed through the loop
//     body and spill (each reload behind a vmcnt(0)); inline-asm MFMAs with "+a" accumulators and "+v" in-place loads
//     give exactly the intended loop in 128 + 100 registers; loop-invariant per-lane values are better kept in LDS;
//   * LDS-DMA and register loads do NOT retire in order with respect to each other (see the waits in the K loop).
//
// Serves the same reference call sites as gg_fwd_patch_k: the Conv2d k4 s2 p1 / ConvTranspose2d k4 s2 p1 layers of
// EncoderBlock / DecoderBlock (models/pix2pix.py:58-111), DiscriminatorBlock 1-3 (models/wrapper.py:229-232) and
// the input-gradient halves of their aten::convolution_backward calls.
#include <stddef.h>

#include "gg_tile.h"

constexpr int BD_CK = 32;                                   // channels per patch chunk = K of one MFMA
constexpr int BD_TH = 16, BD_NW = 4, BD_MT = 8, BD_NT = 4;
constexpr int BD_BM = BD_TH * 16, BD_BN = 128, BD_NTHR = BD_NW * 64;
constexpr int BD_PIX = (BD_TH + 1) * PATCH_W;
constexpr int BD_PPI = BD_NTHR / 4;                         // patch pixels (64 B) per block-wide fill instruction
constexpr int BD_PJ = (BD_PIX + BD_PPI - 1) / BD_PPI;
constexpr int BD_PIXR = (BD_PIX + 15) / 16 * 16;            // a wave's piece of a fill instruction is 16 pixels
constexpr int BD_PBYTES = BD_PIXR * 64;
constexpr int BD_NPB = 3;                                   // patch buffers
constexpr int BD_DUMP = BD_NPB * BD_PBYTES;                 // pieces wholly behind the last patch pixel land here
constexpr int BD_PTAB = BD_DUMP + 1024;                     // [PJ][NTHR] fill map of the patch (kept out of the registers)
constexpr size_t BD_LDS_LOOP = (size_t)BD_PTAB + (size_t)BD_PJ * BD_NTHR * 4;
constexpr size_t BD_LDS_EPI = (size_t)BD_BM * (BD_BN * 2 + 16) + (size_t)BD_NW * 2 * BD_BN * sizeof(float);
constexpr size_t BD_LDS = BD_LDS_LOOP > BD_LDS_EPI ? BD_LDS_LOOP : BD_LDS_EPI;

#ifndef BD_ABL
#define BD_ABL 0            // timing ablations (results WRONG): 1 no patch fill, 2 no MFMA, 4 no fragment reads, 8 no weight loads
#endif

// WIDE: the four waves side by side along the channels, each over all 256 pixels x 32 channels (16 x 2 MFMA tiles)
// instead of 2 x 2 waves of 128 pixels x 64 channels (8 x 4): no two waves fetch the same weights (half the
// vector-memory bytes per FLOP), twice the pixel-fragment reads from LDS.
template <bool WIDE, bool RELU>
__global__ __launch_bounds__(BD_NTHR, 2) void gg_fwd_bd_k(P2Prob g, FwdArgs a) {
    constexpr int MT = WIDE ? 16 : BD_MT, NT = WIDE ? 2 : BD_NT, BM = BD_BM, BN = BD_BN, NTHR = BD_NTHR, PJ = BD_PJ, PPI = BD_PPI;
    constexpr int PBYTES = BD_PBYTES, PIX = BD_PIX, PIXR = BD_PIXR;
    constexpr int WMW = WIDE ? 1 : 2, NW = BD_NW;
    const int mtiles = g.mtiles, ntiles = g.ntiles;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = WIDE ? 0 : wid >> 1, wn = WIDE ? wid : wid & 1;
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int bn = bid % ntiles;
    bid /= ntiles;
    const int ph = bid % g.nphase;
    const int bm = bid / g.nphase;
    const int n0 = bn * BN;
    const int tpi = g.TY * g.TX;
    const int img = bm / tpi, trem = bm - img * tpi;
    const int gy0 = (trem / g.TX) * BD_TH, gx0 = (trem % g.TX) * 16;
    const int groups = g.groups;

    const int gC1 = g.C1, gC2 = g.C2, grelu1 = g.relu1, grelu2 = g.relu2, gD1 = g.D1, gD2 = g.D2;
    bf16_t* y1p = (bf16_t*)a.y1;
    bf16_t* y2p = (bf16_t*)a.y2;
    // per-phase tables, unpacked with scalar shifts (see P2Prob)
    const unsigned wby16 = (unsigned)(g.wby >> (16 * ph)) & 0xffffu, wbx16 = (unsigned)(g.wbx >> (16 * ph)) & 0xffffu;
    const unsigned tpk32 = (unsigned)(((ph & 2) ? g.toff[1] : g.toff[0]) >> (32 * (ph & 1)));   // 8 bits per window
    const unsigned long long wpk = ph == 0 ? g.wt[0] : (ph == 1 ? g.wt[1] : (ph == 2 ? g.wt[2] : g.wt[3]));   // 16 bits per window
    const int poy = (int)((g.poy >> ph) & 1u), pox = (int)((g.pox >> ph) & 1u);
    auto win_by = [&](int q) __attribute__((always_inline)) -> int { return (int)((wby16 >> (4 * q)) & 15u) - 8; };
    auto win_bx = [&](int q) __attribute__((always_inline)) -> int { return (int)((wbx16 >> (4 * q)) & 15u) - 8; };

    // ---- patch fill map: thread -> (pixel p = 64 j + tid / 4, 16-B slot tid % 4); the two 32-B halves of a pixel are
    // swapped when bit 2 of its column px = p % 17 is set (scripts/lds_swizzle_check.py p2_patch: conflict-free)
    const int psc = tid & 3, pl = tid >> 2;
    // bits 0-25 source pixel at window offset (0, 0), 26-27 this lane's 16-B chunk, 28-31 inside the image for window q
    unsigned* ptab = (unsigned*)(smem + BD_PTAB);
#pragma unroll
    for (int j = 0; j < PJ; ++j) {
        const int p = j * PPI + pl;
        const int py = p / PATCH_W, px = p - py * PATCH_W;
        const int y = (gy0 + py) * g.S, x = (gx0 + px) * g.S;
        unsigned m = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int yy = y + win_by(q), xx = x + win_bx(q);
            if (q < groups && p < PIX && (unsigned)yy < (unsigned)g.H && (unsigned)xx < (unsigned)g.W) m |= 1u << q;
        }
        ptab[j * NTHR + tid] = ((unsigned)((img * g.H + y) * g.W + x) & 0x3ffffffu) | ((unsigned)(psc ^ (((px >> 2) & 1) << 1)) << 26) | (m << 28);
    }

    // ---- weights: fragment-major copy behind the row-major pack (pai_pack_frag) ---------------------------------
    // block (64-channel tile t, 32-deep K slice s) = 4 KB at ((t * nsub + s) << 12): [nt][lane][8 bf16]
    const unsigned kelems = (unsigned)(g.wtaps * g.Cin);
    const unsigned wbytes = (unsigned)g.Cout * kelems * 2u;
    // descriptor by hand (base, stride 0, bytes, raw 32-bit data format), wave-uniform: operand of the asm loads below
    const unsigned long long wfa = (unsigned long long)(size_t)a.w + wbytes;
    const u4_t wrs = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)wfa),
                      (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(wfa >> 32)) & 0xffffu,
                      (unsigned)__builtin_amdgcn_readfirstlane((int)wbytes), 0x00020000u};
    const unsigned nsub = kelems >> 5, csub = (unsigned)g.Cin >> 5;
    // WIDE: wave wn takes fragments 2 (wn % 2) + {0, 1} of 64-channel tile wn / 2
    const unsigned wtile = (unsigned)(bn * 2 + (WIDE ? wn >> 1 : wn)) * nsub;
    const unsigned wfrag0 = WIDE ? (unsigned)(wn & 1) * 2048u : 0u;
    const unsigned wv = (unsigned)lane * 16u;

    const int fr = lane & 15, fq = lane >> 4;
    f4_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f4_t){0.f, 0.f, 0.f, 0.f};

    const int cchunks = g.Cin / BD_CK;
    const int gsh = groups == 4 ? 2 : 0;
    const int ngroups = cchunks << gsh;

    // LDS-DMA through buffer descriptors: 32-bit per-lane byte offsets, and a lane whose offset lies beyond the buffer
    // writes ZEROS to LDS (scripts/micro/oob_probe.hip) -- padding pixels need neither a zero line nor 64-bit selects
    // (host, fwd_bd_rows: every tensor is smaller than 2 GB)
    constexpr unsigned OOB = 0x80000000u;
    // (in-image lanes are in range by construction; the descriptors only have to reject the OOB offset)
    const __amdgpu_buffer_rsrc_t x1rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x1), 0, 0x7fffffffu, 0x00020000);
    const __amdgpu_buffer_rsrc_t x2rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x2 ? a.x2 : a.x1), 0, a.x2 ? 0x7fffffffu : 0u, 0x00020000);
#define BD_BLDS16(rs, voff, lptr) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lptr), 16, (int)(voff), 0, 0, 0)
    auto patch_piece = [&](int gi, int b, int j, bool valid) __attribute__((always_inline)) {
        const int c0 = (gi >> gsh) * BD_CK, q = gi & (groups - 1);
        const bool second = c0 >= gC1;
        const int C = second ? gC2 : gC1;
        const int cbase = second ? c0 - gC1 : c0;
        const int dpix = win_by(q) * g.W + win_bx(q);
        if (BD_ABL & 1) return;
        // the fill map lives in LDS (own slots, no barrier needed): five registers the K loop has no room for -- hipcc
        // spilled them and reloaded behind a vmcnt(0) that drained the weight loads; the empty asm keeps it from
        // hoisting the offsets of every (piece, window, tensor) out of the K loop either
        // (the lane id is rematerialised for the same reason: a live tid across the K loop was spilled)
        unsigned ln;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
        unsigned pb = ptab[j * NTHR + wid * 64 + (int)ln];
        asm volatile("" : "+v"(pb));
        const int cofs = cbase + (int)((pb >> 26) & 3u) * 8;
        const unsigned vo = (valid && ((pb >> (28 + q)) & 1u)) ? (unsigned)(((int)(pb & 0x3ffffffu) + dpix) * C + cofs) * 2u : OOB;
        const int p0 = j * PPI + wid * 16;       // first pixel of this wave's piece (wave-uniform)
        unsigned char* dst = smem + ((valid && p0 < PIXR) ? b * PBYTES + p0 * 64 : BD_DUMP);
        if (second) BD_BLDS16(x2rs, vo, dst);
        else BD_BLDS16(x1rs, vo, dst);
    };
    auto fire_patch = [&](int gi, int b, bool valid) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < PJ; ++j) patch_piece(gi, b, j, valid);
    };
    // byte offset of the K slice of (group gi, tap k) in this wave's 64-channel tile
    auto slice_off = [&](int gi, int k) __attribute__((always_inline)) -> unsigned {
        const int q = gi & (groups - 1);
        const unsigned wt = (unsigned)(wpk >> (16 * q + 4 * k)) & 15u;
        return (unsigned)__builtin_amdgcn_readfirstlane((int)(((wtile + wt * csub + (unsigned)(gi >> gsh)) << 12) + wfrag0));
    };
    // LDS address of this lane's fragment piece (patch row wm * MT, tap offset (ty, tx)) in buffer b
    auto patch_base = [&](int b, unsigned t2) __attribute__((always_inline)) -> unsigned {
        const unsigned ty = t2 >> 1, px = (t2 & 1u) + (unsigned)fr;
        return (unsigned)(b * PBYTES) + (((unsigned)(wm * MT) + ty) * PATCH_W + px) * 64u + ((((unsigned)fq) ^ (((px >> 2) & 1u) << 1)) << 4);
    };
    auto tap2 = [&](int gi, int k) __attribute__((always_inline)) -> unsigned {
        return (tpk32 >> (8 * (gi & (groups - 1)) + 2 * k)) & 3u;       // (ty, tx) of the tap
    };
    // ReLU on load as a packed signed-16-bit max against a per-group threshold: 0 (ReLU) or -32768 (identity)
    auto relu_of = [&](int gi) __attribute__((always_inline)) -> int { return ((gi >> gsh) * BD_CK >= gC1 ? grelu2 : grelu1) ? 0 : (int)0x80008000u; };
    auto read_p = [&](unsigned base, int mt) __attribute__((always_inline)) -> bf8_t {
        if (BD_ABL & 4) return __builtin_bit_cast(bf8_t, make_uint4(base, mt, base, mt));
        return *(const bf8_t*)(smem + base + mt * (PATCH_W * 64));
    };
    auto relu_frag_s = [&](bf8_t x, int relu) __attribute__((always_inline)) -> bf8_t {
        typedef __attribute__((ext_vector_type(4))) int i4_t;
        i4_t xi = __builtin_bit_cast(i4_t, x);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int r;
            asm("v_pk_max_i16 %0, %1, %2" : "=v"(r) : "v"(xi[e]), "s"(relu));
            xi[e] = r;
        }
        return __builtin_bit_cast(bf8_t, xi);
    };
    // Weight loads and their waits are inline asm: a builtin load gets fresh registers from hipcc (the K loop then
    // needs > 256 and spills, each reload behind a vmcnt(0) that drains every load in flight); "+v" loads IN PLACE,
    // and the wait carries the register as an operand so that no MFMA using it can be scheduled in front of it.
    // Counts: behind fragment nt of tap s the wave issues 3 - nt + 4 + nt = 7 more weight loads before pass nt of tap
    // s + 2 needs it.
    // NO WAIT MAY COUNT AN LDS-DMA AND A REGISTER LOAD TOGETHER.  vmcnt counts both, but the two kinds do not retire
    // in order WITH RESPECT TO EACH OTHER on gfx950: with "7 + PJ" where a patch fill (PJ LDS-DMA instructions) had been
    // issued between a weight load and its wait, and "2 NT" in front of the group barrier, one wave in ~10^4 read a
    // weight fragment before it had landed -- only on grids of several rounds, where the two workgroups of a CU are
    // out of phase (scripts/micro/bd_race.sh: 37 mismatching runs of 54 against 0 of 54 with the waits below).  So a
    // weight wait is vmcnt(7) whatever lies in between (register loads retire in order among themselves; an LDS-DMA
    // that is still outstanding only makes the wait longer), and "this wave's patch pieces have landed" is vmcnt(0).
    auto load_w1 = [&](u4_t& wf, unsigned soff, int nt) __attribute__((always_inline)) {
        if (BD_ABL & 8) { wf = (u4_t){soff, wv, (unsigned)nt, 1u}; return; }
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "+v"(wf) : "v"(wv), "s"(wrs), "s"(soff + nt * 1024u));
    };
    auto wait_w = [&](u4_t& wf) __attribute__((always_inline)) {
        if (BD_ABL & 8) return;
        asm volatile("s_waitcnt vmcnt(7)" : "+v"(wf));
    };

    // A tap (one 32-deep K slice) is 32 MFMAs per wave, issued CHANNEL-FRAGMENT-MAJOR: for nt: for mt.
    //   weights   two register sets (taps s, s + 1).  Fragment nt of tap s is dead behind pass nt of tap s, and the
    //             same fragment of tap s + 2 is requested into its registers right there: every weight load is two
    //             full taps ahead of its use with only two sets of registers;
    //   pixels    the wave's eight fragments of a tap stay in registers for its four passes (two sets: the fragments of
    //             the next tap are read during the second pass);
    //   patches   three LDS buffers: patch gi + 2 is requested in the last tap of group gi, behind the group's barrier.
    // That ONE barrier per group (128 MFMAs per wave) says (1) patch gi + 1 has landed -- requested a group ago --
    // before the read-ahead crosses into it, and (2) every wave is done with patch gi - 1, whose buffer patch gi + 2
    // overwrites.
    const int S = 4 * ngroups;
    auto slice_of = [&](int s) __attribute__((always_inline)) -> unsigned {
        const int sc = s < S ? s : S - 1;
        return slice_off(sc >> 2, sc & 3);
    };
    if constexpr (WIDE) {
        // Item-major: a tap is 16 items (pixel row mt) of 2 MFMAs; pixel fragments in a ring of eight (the fragment of item
        // i + 8 is read behind the MFMAs of item i), weights in a ring of four sets of two fragments, the set of tap
        // s + 2 requested at the start of tap s.  Counts: behind the set of tap s the wave issues the sets of taps s + 1
        // and s + 2 (4 loads) before it needs it.
        u4_t wB[4][NT];
        bf8_t pf[8];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wB[i][nt] = (u4_t){0u, 0u, 0u, 0u};
        fire_patch(0, 0, true);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) load_w1(wB[0][nt], slice_of(0), nt);
        fire_patch(ngroups > 1 ? 1 : 0, 1, ngroups > 1);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) load_w1(wB[1][nt], slice_of(1), nt);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // patch 0 has landed
        __builtin_amdgcn_s_barrier();
        unsigned pb_cur = patch_base(0, tap2(0, 0));
#pragma unroll
        for (int i = 0; i < 8; ++i) pf[i] = read_p(pb_cur, i);
        int bufc = 0;
        for (int gi = 0; gi < ngroups; ++gi) {
            const int relu = relu_of(gi);
            const bool more = gi + 1 < ngroups;
            const int bufn = bufc == 2 ? 0 : bufc + 1, bufp = bufc == 0 ? 2 : bufc - 1;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k == 3) {
                    if (more) {
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(0) : "memory");   // this wave's pieces of patch gi + 1 have landed
                        __builtin_amdgcn_s_barrier();
                    }
                    fire_patch(gi + 2 < ngroups ? gi + 2 : gi, bufp, gi + 2 < ngroups);
                }
                const unsigned pb_next = k < 3 ? patch_base(bufc, tap2(gi, k + 1)) : (more ? patch_base(bufn, tap2(gi + 1, 0)) : pb_cur);
                const unsigned soff2 = slice_of(4 * gi + k + 2);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) load_w1(wB[(k + 2) & 3][nt], soff2, nt);
                if (!(BD_ABL & 8)) {
                    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(wB[k][0]), "+v"(wB[k][1]) : "n"(2 * NT));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    bf8_t x = pf[mt & 7];
                    if (RELU) x = relu_frag_s(x, relu);
                    if (BD_ABL & 2) acc[mt][0][0] += (float)x[0] + __uint_as_float(wB[k][mt & 1][mt & 3]);
                    else {
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[mt][nt]) : "v"(wB[k][nt]), "v"(x));
                    }
                    pf[mt & 7] = read_p(mt < 8 ? pb_cur : pb_next, mt < 8 ? mt + 8 : mt - 8);
                    __builtin_amdgcn_sched_barrier(0);   // hipcc otherwise sinks every read to its first use
                }
                pb_cur = pb_next;
            }
            bufc = bufn;
        }
    } else {
        u4_t wB[2][NT];
        bf8_t pf[2][MT];
    #pragma unroll
        for (int nt = 0; nt < NT; ++nt) wB[0][nt] = wB[1][nt] = (u4_t){0u, 0u, 0u, 0u};
        // prologue in the order the counts above assume: patch 0, tap 0, patch 1, tap 1
        fire_patch(0, 0, true);
    #pragma unroll
        for (int nt = 0; nt < NT; ++nt) load_w1(wB[0][nt], slice_of(0), nt);
        fire_patch(ngroups > 1 ? 1 : 0, 1, ngroups > 1);
    #pragma unroll
        for (int nt = 0; nt < NT; ++nt) load_w1(wB[1][nt], slice_of(1), nt);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // patch 0 has landed
        __builtin_amdgcn_s_barrier();
        unsigned pb_cur = patch_base(0, tap2(0, 0));
    #pragma unroll
        for (int mt = 0; mt < MT; ++mt) pf[0][mt] = read_p(pb_cur, mt);
        int bufc = 0;
        for (int gi = 0; gi < ngroups; ++gi) {
            const int relu = relu_of(gi);
            const bool more = gi + 1 < ngroups;
            const int bufn = bufc == 2 ? 0 : bufc + 1, bufp = bufc == 0 ? 2 : bufc - 1;
    #pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k == 3) {
                    if (more) {
                        // patch gi + 1 (requested a group ago) has landed: only the weight loads of the last two taps may be outstanding
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of patch gi + 1 have landed
                        __builtin_amdgcn_s_barrier();
                    }
                    fire_patch(gi + 2 < ngroups ? gi + 2 : gi, bufp, gi + 2 < ngroups);
                }
                // (behind the last tap the read-ahead re-reads the current patch: no branch in the MFMA stream)
                const unsigned pb_next = k < 3 ? patch_base(bufc, tap2(gi, k + 1)) : (more ? patch_base(bufn, tap2(gi + 1, 0)) : pb_cur);
                const unsigned soff2 = slice_of(4 * gi + k + 2);
                __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    wait_w(wB[k & 1][nt]);
    #pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        if (RELU && nt == 0) pf[k & 1][mt] = relu_frag_s(pf[k & 1][mt], relu);
                        if (BD_ABL & 2) acc[mt][nt][0] += (float)pf[k & 1][mt][0] + __uint_as_float(wB[k & 1][nt][mt & 3]);
                        else
                            // D[i = channel slot][j = pixel]: acc[mt][nt][r] = channel slot 4 fq + r of pixel fr.  asm with the
                            // accumulator tied in place in the ACCUMULATION registers ("+a"): as a builtin hipcc 7.2 renames the
                            // 128 accumulator registers through the loop body and spills
                            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[mt][nt]) : "v"(wB[k & 1][nt]), "v"(pf[k & 1][mt]));
                        // the next tap's fragments, one read behind every third MFMA of the first three passes: spread
                        // evenly the 12 other vector-memory / LDS instructions of a tap cost the matrix pipe least
                        // (scripts/micro/mfma_mix.hip: 1.64 PFLOP/s against 1.43 with the eight reads behind consecutive MFMAs)
                        if ((nt * MT + mt) % 3 == 2 && (nt * MT + mt) / 3 < MT) pf[(k & 1) ^ 1][(nt * MT + mt) / 3] = read_p(pb_next, (nt * MT + mt) / 3);
                        if ((nt * MT + mt) % 3 == 2 || (RELU && nt == 0)) __builtin_amdgcn_sched_barrier(0);   // hipcc otherwise sinks every read to its first use
                    }
                    load_w1(wB[k & 1][nt], soff2, nt);
                    __builtin_amdgcn_sched_barrier(0);
                }
                pb_cur = pb_next;
            }
            bufc = bufn;
        }
    }
    // (the compiler does not know the asm above are MFMAs: their results are read after the pipeline has drained)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    __builtin_amdgcn_s_barrier();   // the epilogue reuses the patch memory

    // ---- epilogue: bias, BN partial statistics, activation, LDS-staged row stores (as gg_fwd_patch_k) ----------
    constexpr int CROW = BN * 2 + 16;
    unsigned char* Cs = smem;
    float* sstat = (float*)(smem + BM * CROW);  // [WMW][2][BN] (forward) / [NW][2][BN] (bwd_write_partials)
    const int eact = a.yact ? a.eact : PAI_ACT_NONE;
    // lane (fq, fr) holds, for each of its MT pixel rows mt (pixel fr of the row), the 16 consecutive channels
    // wn*64 + 16 fq + (4 nt + r)
    constexpr int CL = 4 * NT;
    // (WIDE: the 8 channels 16 fq + 8 (wn % 2) + (4 nt + r) of tile wn / 2)
    const int col0 = WIDE ? (wn >> 1) * 64 + 16 * fq + 8 * (wn & 1) : wn * 64 + CL * fq;
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

// ---- fragment-major copy of a row-major bf16 filter pack --------------------------------------------------------
// dst block (64-row tile t, 32-deep K slice s) at ((t * K/32 + s) * 2048) elements: [nt 4][lane 64][8 elements], lane
// (fq = lane / 16, fr = lane % 16) of fragment nt = row 64 t + 16 (fr / 4) + 4 nt + fr % 4, elements 32 s + 8 fq .. + 8:
// the MFMA operand a lane of gg_fwd_bd_k holds (16 CONSECUTIVE channels of one pixel per lane in its epilogue).
__global__ __launch_bounds__(256) void pack_frag_k(const uint4* src, int rows, int K, uint4* dst) {
    const int nsub = K >> 5;
    const int64_t n = (int64_t)rows * K / 8;
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < n; o += (int64_t)gridDim.x * 256) {
        const int lane = (int)(o & 63), nt = (int)((o >> 6) & 3);
        const int64_t blk = o >> 8;
        const int s = (int)(blk % nsub), t = (int)(blk / nsub);
        const int fr = lane & 15, fq = lane >> 4;
        const int row = t * 64 + 16 * (fr >> 2) + 4 * nt + (fr & 3);
        dst[o] = src[((int64_t)row * K + 32 * s + 8 * fq) >> 3];
    }
}

extern "C" int pai_pack_frag(const void* w_rowmajor, int rows, int K, void* w_frag, void* stream) {
    PAI_CHECK(w_rowmajor && w_frag && rows > 0 && K > 0, "pai_pack_frag: bad arguments");
    PAI_CHECK(rows % 64 == 0 && K % 32 == 0, "pai_pack_frag: rows=%d must be a multiple of 64, K=%d of 32", rows, K);
    const int64_t n = (int64_t)rows * K / 8;
    const int64_t blocks = (n + 255) / 256;
    PAI_LAUNCH(pack_frag_k, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, (hipStream_t)stream,
                       (const uint4*)w_rowmajor, rows, K, (uint4*)w_frag);
    PAI_LAUNCH_CHECK();
    return 0;
}

// ---- host side --------------------------------------------------------------------------------------------
// tunable "fwd_bd": 0 off, 1 the 2 x 2-wave form, 2 the four-waves-side-by-side form, for every problem it accepts; "fwd_bd_min_wgs": smallest grid it takes
int fwd_bd_rows(const GG& g) {
    if (!g.wfrag || !pai_tunable("fwd_bd", 1)) return 0;
    if ((g.C1 % BD_CK) || (g.C2 % BD_CK) || g.Cin < 64) return 0;
    // packed pixel indices; 32-bit byte offsets into buffer descriptors
    if (((int64_t)g.N * g.H + 4) * (g.W + 4) >= (1 << 26) || (int64_t)g.Cout * g.wtaps * g.Cin * 2 >= (1ll << 31)) return 0;
    if ((int64_t)g.N * g.H * g.W * (g.C1 > g.C2 ? g.C1 : g.C2) * 2 >= (1ll << 31)) return 0;
    if ((g.Cout % 128) || (g.D2 != 0 && (g.D1 % 128))) return 0;
    PatchGeo pg;
    if (!patch_geo(g, BD_TH, &pg)) return 0;
    if ((int64_t)(g.M / BD_BM) * (g.Cout / BD_BN) * g.nphase < pai_tunable("fwd_bd_min_wgs", 512)) return 0;
    return BD_BM;
}

int launch_fwd_bd(const GG& g, const FwdArgs& a, hipStream_t s) {
    PatchGeo pg;
    PAI_CHECK(fwd_bd_rows(g) && patch_geo(g, BD_TH, &pg), "launch_fwd_bd: problem not eligible");
    static bool attr = false;
    if (!attr) {
        const void* fns[4] = {reinterpret_cast<const void*>(&gg_fwd_bd_k<false, false>), reinterpret_cast<const void*>(&gg_fwd_bd_k<false, true>),
                              reinterpret_cast<const void*>(&gg_fwd_bd_k<true, false>), reinterpret_cast<const void*>(&gg_fwd_bd_k<true, true>)};
        for (int i = 0; i < 4; ++i) {
            hipError_t e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)BD_LDS);
            PAI_CHECK(e == hipSuccess, "hipFuncSetAttribute(max dynamic LDS): %s", hipGetErrorString(e));
        }
        attr = true;
    }
    const int mtiles = g.M / BD_BM, ntiles = g.Cout / BD_BN;
    P2Prob pr;
    p2_prob(g, pg, mtiles, ntiles, &pr);
    const dim3 grid(mtiles * ntiles * g.nphase), block(BD_NTHR);
    const bool relu = g.relu1 || g.relu2;
    if (pai_tunable("fwd_bd", 1) == 2) {
        if (relu) PAI_LAUNCH((gg_fwd_bd_k<true, true>), grid, block, BD_LDS, s, pr, a);
        else PAI_LAUNCH((gg_fwd_bd_k<true, false>), grid, block, BD_LDS, s, pr, a);
    } else {
        if (relu) PAI_LAUNCH((gg_fwd_bd_k<false, true>), grid, block, BD_LDS, s, pr, a);
        else PAI_LAUNCH((gg_fwd_bd_k<false, false>), grid, block, BD_LDS, s, pr, a);
    }
    PAI_LAUNCH_CHECK();
    return 0;
}

const char* fwd_bd_kernel_name(const GG& g) {
    const bool relu = g.relu1 || g.relu2;
    if (pai_tunable("fwd_bd", 1) == 2) return relu ? "gg_fwd_bd_k<true, true>" : "gg_fwd_bd_k<true, false>";
    return relu ? "gg_fwd_bd_k<false, true>" : "gg_fwd_bd_k<false, false>";
}

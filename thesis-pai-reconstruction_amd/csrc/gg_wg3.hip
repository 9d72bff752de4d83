// Patch-resident weight gradient with 128 x 64 WAVE tiles ("wgrad3"):
//   dW[co][tap][ci] = sum_pixels dY[pix][co] * X[pix + tap][ci]
//
// What bounds gg_wgrad_patch_k (gg_mfma.hip) is the LDS port, not the matrix pipe and not the fill volume (round 2:
// fills alone and MFMAs alone take ~110 us each on decoders[4] and do not overlap; gg_wg2.hip halved the fill per FLOP
// with the SAME 64 x 64 wave tile and was not faster): per 64-pixel step its four 64 x 64 waves read 64 KB of
// transposed fragments and receive 21 KB of LDS-DMA for 512 matrix cycles per SIMD = 166 B per cycle of a port that
// peaks at 256.  The lever is LDS bytes per FLOP, and only the WAVE tile changes the fragment share of it.
// Here a workgroup is four waves side by side along the columns; wave t owns tap t of the 2 x 2 window x CI input
// channels and ALL BMC output channels of the tile:
//   BMC = 128, CI = 64:  wave tile 128 x 64 (8 + 4 fragments per 32 MFMAs instead of 4 + 4 per 16), workgroup tile
//                        128 x 256.  Per 64-pixel step: 96 KB of fragment reads + 27 KB of fills for 1024 matrix
//                        cycles per SIMD = 120 B per cycle (-28 %).
//   BMC = 64, CI = 128:  wave tile 64 x 128 (4 + 8 fragments per 32 MFMAs), workgroup tile 64 x 512: the layers with 64
//                        output channels (decoders[6]) read dY once per 128 input channels instead of once per 32 --
//                        gg_wgrad_patch_k<64> moved 820 MB of HBM traffic for 268 MB of operands there.
// 128 accumulator registers per lane, two workgroups per CU, two LDS stages (the tiles of step k + 1 are in flight
// while step k is multiplied, one barrier per step), LDS-DMA through buffer descriptors as in gg_wgrad_patch_k.
//
// Pixel splits no longer ADD their partial tiles with fp32 atomics (the memory-side atomic rate is ~1.3 TB/s and one
// round of workgroups flushes 32-64 MB at the same moment: 19 % of gg_wgrad_patch_k): every split stores its tile into
// its own slab of the handle's weight-gradient workspace with plain stores and wgrad_slab_sum_k adds the slabs in
// split order (deterministic) into dW.
//
// Serves the weight-gradient half of aten::convolution_backward of the dense Conv2d / ConvTranspose2d k4 s2 p1 layers
// (reference models/pix2pix.py:58-111, models/wrapper.py:229-232).
#include <type_traits>

#include "gg_tile.h"

__device__ __forceinline__ int tr_swz3(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ unsigned tr_off3(int row, int ch) { return (unsigned)(256 * row + 16 * (ch ^ tr_swz3(row))); }
// X patch image: pixel p = py * 17 + px at byte XPB * p (XPB = 2 CI bytes); its 32-B segments (16 channels) are stored
// at segment s ^ f(p), f(p) = bit 1 of p | bit 3 of p << 1 (gg_wg2.hip): the eight pixel rows a half-wave of a
// ds_read_b64_tr_b16 touches (p0 .. p0+3 and p0+8 .. p0+11, any p0) then cover the 64 banks once, for 128-B and for
// 256-B pixels alike (scripts/lds_swizzle_check.py wg3_x)
//   128-B pixels (CI = 64):  4 segments, bank half = bit 0 of p, f(p) = bit 1 | bit 3 << 1
//   256-B pixels (CI = 128): 8 segments,                        f(p) = bit 0 | bit 1 << 1 | bit 3 << 2
//   BW = 8 (8 x 8-pixel K steps: the 8 x 8 images of encoders[4] / decoders[3]): patch rows of 9 pixels at a pitch of
//   12, the half-wave's rows are p0 .. p0+3 and p0+12 .. p0+15: f(p) = bit 1 | bit 2 << 1 (128-B pixels; pitches 9, 10
//   and 17 admit no XOR swizzle at all -- scripts/lds_swizzle_check.py wg3_x8)
template <int CI, int BW> __device__ __forceinline__ unsigned xseg_swz3(unsigned p) {
    if (BW == 8) return ((p >> 1) & 1u) | (((p >> 2) & 1u) << 1);
    return CI == 64 ? (((p >> 1) & 1u) | (((p >> 3) & 1u) << 1)) : ((p & 3u) | (((p >> 3) & 1u) << 2));
}
// dY tile image: K row r (64 per step) x BMC channels, 16-B chunk c of the row stored at slot c ^ yswz(r).  The 16 rows
// a ds_read_b64_tr_b16 touches per 32-lane half ({0-3, 8-11} + 4 h of a 32-row half step, 32 B each) must cover the
// 64 banks once:
//   256-B rows (BMC = 128): the guide's T10 layout (b), as gg_wgrad_mfma_k
//   128-B rows (BMC = 64):  bank half = bit 0 of r, 32-B segment ^ (bit 1 | bit 3 << 1)
template <int BMC> __device__ __forceinline__ int yswz3(int row) {
    return BMC == 128 ? tr_swz3(row) : ((((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1);
}

#ifndef WG3_PRIO
#define WG3_PRIO 0      // static s_setprio of the weight-gradient waves (1-3 measured in the step: 5.70-5.73 against 5.66-5.70 ms -- not used)
#endif
#ifndef WG3_SPREAD
#define WG3_SPREAD 0   // 1: the next step's LDS-DMA pieces between the MFMAs of k-half 0 instead of in one block in front of them (measured: 1964-1970 against 1917-1947 us over the wgrad layers, step 5.82 against 5.79 ms -- dropped)
#endif
#ifndef WG3_ADDR
#define WG3_ADDR 1    // pipelined loop: 1 = one address register per fragment tile (32 registers), 0 = base ^ tile index in front of each read
#endif
#ifndef WG3_ABL
#define WG3_ABL 0     // timing ablations (results WRONG): 1 no dW store, 2 no MFMA, 4 no fills, 8 no fragment reads
#endif

// ---- pinned fragment registers (round 6) ------------------------------------------------------------------------------
// The two 8-byte halves of an MFMA operand come from two ds_read_b64_tr_b16 (inline asm: the compiler must not see the
// reads, see the K loop).  With "=v" outputs the allocator does not place a fragment's halves side by side (an inline-asm
// definition is not coalesced into a sub-register of the operand tuple) and `compose` becomes a v_mov_b64 per half.
// Pinned fragments live in FIXED registers v[160 + 4 F .. 163 + 4 F] (F = 12 k-half + fragment): the reads deliver into
// the halves in place and the composition is free.
//   round-3 loop: measured with ALL 24 fragments pinned and not used (32 -> 0 v_mov_b64 per 64 MFMAs, launches 1.6 % slower:
//                 vector instructions per MFMA are not what bounds a loop whose matrix pipe idles behind every barrier).
//   pipelined loop (WG3_PIN, default 1): the dY fragments are pinned; the X fragments go through the ReLU (max(x, rlo) on
//                 the halves: compiler-generated, so ITS results are placed in the tuple) and need no copy either.
#ifndef WG3_PIN
#define WG3_PIN 1
#endif
typedef __attribute__((ext_vector_type(2))) unsigned wg3_u2_t;
#define WG3_PIN_CASES(X) X(0, 160, 161, 162, 163) X(1, 164, 165, 166, 167) X(2, 168, 169, 170, 171) X(3, 172, 173, 174, 175) X(4, 176, 177, 178, 179) X(5, 180, 181, 182, 183) X(6, 184, 185, 186, 187) X(7, 188, 189, 190, 191) X(8, 192, 193, 194, 195) X(9, 196, 197, 198, 199) X(10, 200, 201, 202, 203) X(11, 204, 205, 206, 207) X(12, 208, 209, 210, 211) X(13, 212, 213, 214, 215) X(14, 216, 217, 218, 219) X(15, 220, 221, 222, 223) X(16, 224, 225, 226, 227) X(17, 228, 229, 230, 231) X(18, 232, 233, 234, 235) X(19, 236, 237, 238, 239) X(20, 240, 241, 242, 243) X(21, 244, 245, 246, 247) X(22, 248, 249, 250, 251) X(23, 252, 253, 254, 255)
template <int F, unsigned OFF, bool PINNED>
__device__ __forceinline__ void wg3_rd_pair(wg3_u2_t& lo, wg3_u2_t& hi, unsigned o0, unsigned o1) {
    if constexpr (PINNED) {
#define X(F_, R0, R1, R2, R3)                                                                                          \
    if constexpr (F == F_) {                                                                                           \
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "={v[" #R0 ":" #R1 "]}"(lo) : "v"(o0), "n"(OFF));          \
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "={v[" #R2 ":" #R3 "]}"(hi) : "v"(o1), "n"(OFF));          \
    }
        WG3_PIN_CASES(X)
#undef X
    } else {
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(o0), "n"(OFF));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(o1), "n"(OFF));
    }
}
// The pipelined loop's forms: ONE asm statement per fragment -- address XORs (tile index, see the fragment addresses) and
// both reads; between separate statements hipcc pads with an s_nop each (it cannot see what they do), and every instruction of
// a wave that is alone on its SIMD is an issue slot.  XI: the tile's XOR immediate (0: none).
template <unsigned OFF>
__device__ __forceinline__ void wg3_rdx_pair(wg3_u2_t& lo, wg3_u2_t& hi, unsigned b0, unsigned b1, unsigned xi) {
    unsigned t0, t1;
    asm volatile("v_xor_b32 %2, %6, %4\n\tv_xor_b32 %3, %6, %5\n\tds_read_b64_tr_b16 %0, %2 offset:%7\n\tds_read_b64_tr_b16 %1, %3 offset:%7"
                 : "=&v"(lo), "=&v"(hi), "=&v"(t0), "=&v"(t1) : "v"(b0), "v"(b1), "s"(xi), "n"(OFF));
}
// ... and with the address of every tile in a register of its own (WG3_ADDR): two reads, nothing else -- the XOR in front of a
// read is a vector instruction the read depends on, and a lone wave pays ~9 matrix cycles for every instruction between MFMAs
// (the reads and their XORs compiled out: decoders[4] 157 -> 108 us)
template <unsigned OFF>
__device__ __forceinline__ void wg3_rd2(wg3_u2_t& lo, wg3_u2_t& hi, unsigned a0, unsigned a1) {
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%4\n\tds_read_b64_tr_b16 %1, %3 offset:%4" : "=&v"(lo), "=&v"(hi) : "v"(a0), "v"(a1), "n"(OFF));
}
template <int F, unsigned OFF>
__device__ __forceinline__ void wg3_rd2_pinned(wg3_u2_t& lo, wg3_u2_t& hi, unsigned a0, unsigned a1) {
#define X(F_, R0, R1, R2, R3)                                                                                          \
    if constexpr (F == F_)                                                                                             \
        asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%4\n\tds_read_b64_tr_b16 %1, %3 offset:%4"                      \
                     : "={v[" #R0 ":" #R1 "]}"(lo), "={v[" #R2 ":" #R3 "]}"(hi) : "v"(a0), "v"(a1), "n"(OFF));
    WG3_PIN_CASES(X)
#undef X
}
template <int F, unsigned OFF>
__device__ __forceinline__ void wg3_rdx_pair_pinned(wg3_u2_t& lo, wg3_u2_t& hi, unsigned b0, unsigned b1, unsigned xi) {
    unsigned t0, t1;
#define X(F_, R0, R1, R2, R3)                                                                                          \
    if constexpr (F == F_)                                                                                             \
        asm volatile("v_xor_b32 %2, %6, %4\n\tv_xor_b32 %3, %6, %5\n\tds_read_b64_tr_b16 %0, %2 offset:%7\n\tds_read_b64_tr_b16 %1, %3 offset:%7" \
                     : "={v[" #R0 ":" #R1 "]}"(lo), "={v[" #R2 ":" #R3 "]}"(hi), "=&v"(t0), "=&v"(t1) : "v"(b0), "v"(b1), "s"(xi), "n"(OFF));
    WG3_PIN_CASES(X)
#undef X
}
// MFMA whose A operand is the pinned slot F: the register tuple is NAMED in the instruction, its halves are inputs pinned to the
// same registers (no copy, no composition, and the compiler still sees who reads what the fragment reads deliver)
template <int F>
__device__ __forceinline__ void wg3_mfma_pinned(f4_t& acc, wg3_u2_t lo, wg3_u2_t hi, bf8_t b) {
#define X(F_, R0, R1, R2, R3)                                                                                          \
    if constexpr (F == F_)                                                                                             \
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, v[" #R0 ":" #R3 "], %1, %0" : "+v"(acc) : "v"(b), "{v[" #R0 ":" #R1 "]}"(lo), "{v[" #R2 ":" #R3 "]}"(hi));
    WG3_PIN_CASES(X)
#undef X
}
// ties the halves (and then the composed operand) to this point of the program, in their registers
template <int F, bool PINNED>
__device__ __forceinline__ bf8_t wg3_compose(wg3_u2_t lo, wg3_u2_t hi) {
    if constexpr (PINNED) {
#define X(F_, R0, R1, R2, R3)                                                                                          \
    if constexpr (F == F_) asm volatile("" : "={v[" #R0 ":" #R1 "]}"(lo), "={v[" #R2 ":" #R3 "]}"(hi) : "0"(lo), "1"(hi));
        WG3_PIN_CASES(X)
#undef X
        bf8_t f = __builtin_bit_cast(bf8_t, make_uint4(lo.x, lo.y, hi.x, hi.y));
#define X(F_, R0, R1, R2, R3) \
    if constexpr (F == F_) asm volatile("" : "={v[" #R0 ":" #R3 "]}"(f) : "0"(f));
        WG3_PIN_CASES(X)
#undef X
        return f;
    } else {
        asm volatile("" : "+v"(lo), "+v"(hi));
        return __builtin_bit_cast(bf8_t, make_uint4(lo.x, lo.y, hi.x, hi.y));
    }
}
// the same with the B operand in the pinned slot (64 x 128 wave tile: the X fragments are the long side)
template <int F>
__device__ __forceinline__ void wg3_mfma_pinned_b(f4_t& acc, bf8_t a, wg3_u2_t lo, wg3_u2_t hi) {
#define X(F_, R0, R1, R2, R3)                                                                                          \
    if constexpr (F == F_)                                                                                             \
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, v[" #R0 ":" #R3 "], %0" : "+v"(acc) : "v"(a), "{v[" #R0 ":" #R1 "]}"(lo), "{v[" #R2 ":" #R3 "]}"(hi));
    WG3_PIN_CASES(X)
#undef X
}
// ReLU of a pinned fragment in place: max(x, rlo) on signed 16-bit lanes, rlo2 = two copies of 0 or of -32768 (no ReLU)
template <int F>
__device__ __forceinline__ void wg3_relu_pinned(wg3_u2_t& lo, wg3_u2_t& hi, unsigned rlo2) {
#define X(F_, R0, R1, R2, R3)                                                                                          \
    if constexpr (F == F_)                                                                                             \
        asm volatile("v_pk_max_i16 v" #R0 ", v" #R0 ", %2\n\tv_pk_max_i16 v" #R1 ", v" #R1 ", %2\n\tv_pk_max_i16 v" #R2 ", v" #R2 ", %2\n\tv_pk_max_i16 v" #R3 ", v" #R3 ", %2" \
                     : "={v[" #R0 ":" #R1 "]}"(lo), "={v[" #R2 ":" #R3 "]}"(hi) : "s"(rlo2), "0"(lo), "1"(hi));
    WG3_PIN_CASES(X)
#undef X
}
__device__ __forceinline__ void wg3_tie(wg3_u2_t& lo, wg3_u2_t& hi) { asm volatile("" : "+v"(lo), "+v"(hi)); }
template <int N, typename FN, int... I>
__device__ __forceinline__ void wg3_sfor_impl(FN&& fn, std::integer_sequence<int, I...>) { (fn(std::integral_constant<int, I>{}), ...); }
template <int N, typename FN>
__device__ __forceinline__ void wg3_sfor(FN&& fn) { wg3_sfor_impl<N>(fn, std::make_integer_sequence<int, N>{}); }

// BW: width of the K step's pixel block: 16 (4 x 16 pixels, images >= 16 wide) or 8 (8 x 8 pixels: 8 x 8 images)
// PIPE: 1 = the pipelined K loop (round 6, see there), 0 = the round-3 loop (tunable wgrad3_pipe).  Both keep TWO LDS stages;
// a third one (the fills of steps k + 1 AND k + 2 in flight, 84-96 KB per workgroup) was built and measured in round 6: every
// layer within +-2 %, step 5.71-5.74 against 5.73-5.74 ms -- removed.
template <int BMC, int CI, int BW = 16, int PIPE = 1>
__global__ __launch_bounds__(256, 2) void gg_wgrad_patch3_k(GG g, WgradArgs a, PatchGeo pg, int cotiles, int jtiles,
                                                            int splits, int blocks_per_split, int ph_inner) {
    static_assert((BMC == 128 && CI == 64) || (BMC == 64 && CI == 128), "wave tile 128 x 64 or 64 x 128");
    static_assert(BW == 16 || (BW == 8 && CI == 64), "8-wide blocks: 128-B patch pixels only");
    constexpr int LBW = BW == 16 ? 4 : 3;        // log2 BW
    constexpr int BH = 64 / BW, LBH = BW == 16 ? 2 : 3;   // rows of the block
    constexpr int PW = BW == 16 ? PATCH_W : 12;  // pitch of a patch row in pixels (BW + 1 used)
    constexpr int MT = BMC / 16;                 // 16-row MFMA tiles along the output channels
    constexpr int NT = CI / 16;                  // 16-column MFMA tiles along the input channels
    constexpr int YROW = BMC * 2;                // bytes per pixel row of the dY tile (256 or 128)
    constexpr int YBUF = 64 * YROW;
    constexpr int XPB = CI * 2;                  // bytes per patch pixel
    constexpr int XSLOTS = BW == 16 ? 96 : 128;  // pixel slots filled (5 x 17 = 85 / 9 rows at pitch 12 = 108 used)
    constexpr int XBUF = XSLOTS * XPB;
    constexpr int STAGE = YBUF + XBUF;
    constexpr int YJ = YBUF / 4096;              // dY fill instructions per thread (256 threads x 16 B each)
    constexpr int XJ = XBUF / 4096;              // X fill instructions per thread
    constexpr int XCH = XPB / 16;                // 16-B chunks per patch pixel
    constexpr int XPPI = 256 / XCH;              // patch pixels per block-wide fill instruction
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    if (WG3_PRIO) __builtin_amdgcn_s_setprio(WG3_PRIO);         // static issue priority of the weight-gradient waves (see WG3_PRIO)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // = tap of the window
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    // transposed layers: the four output phases of one (column tile, pixel range) read the SAME X patches -- as neighbours in
    // the tile order they run at the same time on one XCD and three of the four reads are L2 hits (decoders[6]: 804 MB of
    // HBM traffic for 268 MB of operands with the phase outermost)
    int ph = 0;
    if (ph_inner) { ph = bid % g.nphase; bid /= g.nphase; }
    const int jt = bid % jtiles; bid /= jtiles;
    const int cot = bid % cotiles; bid /= cotiles;
    const int split = bid % splits;
    if (!ph_inner) ph = bid / splits;
    const int co0 = cot * BMC;
    const int q = jt & (pg.groups - 1);
    const int ci0 = (jt >> (pg.groups == 4 ? 2 : 0)) * CI;

    // CI-channel column tiles never straddle the two sources (host: C1 % CI == 0)
    const bool second = ci0 >= g.C1;
    const bf16_t* xsrc = second ? (const bf16_t*)a.x2 : (const bf16_t*)a.x1;
    const int xcs = second ? g.C2 : g.C1;
    const int xrelu = second ? g.relu2 : g.relu1;
    // (an offset of 0x80000000 is beyond every buffer: the LDS-DMA writes zeros)
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(a.dy), 0, (unsigned)((g.N << (g.ldh + g.ldw)) * g.Cout) * 2u, 0x00020000);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>((const void*)xsrc), 0, (unsigned)(g.N * g.H * g.W * xcs) * 2u, 0x00020000);

    const int los = g.OS == 2 ? 1 : 0;
    const int poy = g.poy[ph], pox = g.pox[ph];
    // dY tile fill map: K row r of the step = pixel (gy0 + (r >> 4), gx0 + (r & 15)) of its 4 x 16 block.
    //   BMC = 128: 256-B rows, 16 chunks; thread -> (row 16 j + tid / 16, slot tid % 16), slot holds chunk slot ^ swz(row)
    //   BMC = 64:  128-B rows,  8 chunks; thread -> (row 32 j + tid / 8,  slot tid % 8)
    constexpr int YCH = YROW / 16, YRPI = 256 / YCH;   // chunks per row, rows per block-wide fill instruction
    const int ysr = tid / YCH, ysc = tid % YCH;
    const int ygch = ysc ^ yswz3<BMC>(ysr);      // the swizzle only looks at row bits 0-3: the same for every j
    const unsigned ythr = (unsigned)(((((ysr >> LBW) << los) << g.ldw) + ((ysr & (BW - 1)) << los)) * g.Cout + co0 + ygch * 8) * 2u;
    // YRPI rows = YRPI / BW pixel rows further.  readfirstlane: hipcc kept this uniform value in a VECTOR register in the 64 x 128
    // form and put a waterfall loop around the second dY piece of every fill -- a branch in the middle of the MFMA sequence,
    // and a block boundary across which it copied fragment registers the reads had not delivered into yet (see WG3_PIN)
    const unsigned yjstep = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(((((YRPI >> LBW) << los) << g.ldw)) * g.Cout) * 2u));
    // X patch fill map: thread -> (pixel XPPI jj + tid / XCH, 16-B chunk tid % XCH)
    const int wby = pg.by[ph][q], wbx = pg.bx[ph][q];
    // (row, column) of the thread's patch pixel as two 16-bit fields: the in-image test of a step is one packed add of the
    // block origin, one packed unsigned min against (H - 1, W - 1) and one compare (was: two adds, two compares, an and; and
    // two registers per piece in a loop that has none to spare).  Rows beyond the patch carry 0x7fff: no image has them.
    typedef unsigned short us2_t __attribute__((ext_vector_type(2)));
    us2_t xyt[XJ];
    unsigned xthr[XJ];
#pragma unroll
    for (int jj = 0; jj < XJ; ++jj) {
        const int p = jj * XPPI + tid / XCH;
        const int py_ = p / PW, px_ = p - py_ * PW;
        xyt[jj] = (us2_t){(unsigned short)((py_ > BH || px_ > BW) ? 0x7fff : py_ * g.S + wby), (unsigned short)(px_ * g.S + wbx)};
        const int c = tid % XCH;                 // physical chunk: segment c >> 1 holds source segment (c >> 1) ^ f(p)
        const int xch = (second ? ci0 - g.C1 : ci0) + (((((c >> 1) ^ (int)xseg_swz3<CI, BW>((unsigned)p)) << 1) | (c & 1)) * 8);
        xthr[jj] = (unsigned)(((py_ * g.S + wby) * g.W + px_ * g.S + wbx) * xcs + xch) * 2u;
    }

    const int lbx = g.lw - LBW, lby = g.lh - LBH;
    const int kb0 = split * blocks_per_split;
    const int kb1 = min(g.M >> 6, kb0 + blocks_per_split);

    const int fi = lane & 15, fg = lane >> 4;
    const int tq = fi >> 2, tp = fi & 3;
    // fragment addresses (integer LDS addresses: the dynamic LDS block starts at 0, checked below)
    //   dY: K row rl = fg * 8 + tq (+ 4 h) of a 32-row half step, channel tile mt: chunk (2 mt + (tp >> 1)) ^ swz(rl), the tile
    //       index is an XOR of address bits 5.. (the swizzle only touches the chunk bits)
    //   X:  patch pixel of K row r + tap shift, segment nt ^ f(p): the tile index is an XOR of address bits 5..
    const unsigned toff = (unsigned)((wid >> 1) * PW + (wid & 1));     // tap slot k sits at patch offset (k >> 1, k & 1) (patch_geo)
    unsigned ybase[2], xbase[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int rowl = fg * 8 + tq + 4 * h;
        ybase[h] = (unsigned)(YROW * rowl + 16 * ((tp >> 1) ^ yswz3<BMC>(rowl)) + 8 * (tp & 1));
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int r = kk * 32 + fg * 8 + tq + 4 * h;
            const unsigned p = (unsigned)((r >> LBW) * PW + (r & (BW - 1))) + toff;
            xbase[kk][h] = (unsigned)YBUF + p * XPB + (xseg_swz3<CI, BW>(p) << 5) + tp * 8;
        }
    }
#define WG3_XOR(dst, src, imm) asm volatile("v_xor_b32 %0, %2, %1" : "=v"(dst) : "v"(src), "s"(imm))
#define WG3_TR(addr) __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf4_t __attribute__((address_space(3)))*)(size_t)(unsigned)(addr))
#define WG3_BLDS16(rs, voff, soff, laddr) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + (laddr)), 16, (int)(voff), (int)(soff), 0, 0)
    if ((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem != 0u) __builtin_trap();

    f4_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f4_t){0.f, 0.f, 0.f, 0.f};

    // Bias gradient = column sums of dY.  Every wave holds the dY fragments of its tile anyway: wave w adds up the
    // fragments of channel tiles w * MT/4 .. of the steps with kb % jtiles == jt, so the jtiles workgroups that read the
    // same dY share the work (a few vector instructions per step each) and no workgroup is slower than the others.
    constexpr int BT = MT / 4;
    const bool do_bias = !PIPE && a.dbias != nullptr;      // the pipelined loop carries no bias sums (host: launch_wgrad3)
    float bsum[BT];
#pragma unroll
    for (int i = 0; i < BT; ++i) bsum[i] = 0.f;

    // addresses of a step's fill, in pieces (the pipelined loop spreads them over the gaps between MFMAs)
    unsigned ysof = 0, xofs[XJ], p_xsof = 0;
    int p_gx0 = 0, p_gy0 = 0, p_n = 0;
    us2_t p_oyx = {0, 0};
    const us2_t xylim = {(unsigned short)(g.H - 1), (unsigned short)(g.W - 1)};
    auto prep_a = [&](int kb) {
        p_gx0 = (kb & ((1 << lbx) - 1)) << LBW;
        p_gy0 = ((kb >> lbx) & ((1 << lby) - 1)) << LBH;
        p_n = kb >> (lbx + lby);
    };
    auto prep_b = [&]() { ysof = (unsigned)(((((p_n << g.ldh) + (p_gy0 << los) + poy) << g.ldw) + (p_gx0 << los) + pox) * g.Cout) * 2u; };
    auto prep_c = [&]() {
        const int oy = p_gy0 * g.S, ox = p_gx0 * g.S;
        p_oyx = (us2_t){(unsigned short)oy, (unsigned short)ox};
        p_xsof = (unsigned)(((((p_n << g.lsh) + oy) << g.lsw) + ox) * xcs) * 2u;
    };
    // (asm: hipcc keeps the limits and the out-of-range offset in vector registers of their own otherwise -- the loop has none)
    const unsigned xylim_s = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(unsigned, xylim));
    auto prep_x = [&](int jj) {
        unsigned t, m, oob;
        asm("v_pk_add_u16 %0, %1, %2" : "=v"(t) : "v"(__builtin_bit_cast(unsigned, xyt[jj])), "s"(__builtin_bit_cast(unsigned, p_oyx)));
        asm("v_pk_min_u16 %0, %1, %2" : "=v"(m) : "v"(t), "s"(xylim_s));
        asm("v_bfrev_b32 %0, 1" : "=v"(oob));        // 0x80000000 = OOB
        xofs[jj] = m == t ? xthr[jj] + p_xsof : oob;
    };
    auto prepare = [&](int kb) {
        prep_a(kb);
        prep_b();
        prep_c();
#pragma unroll
        for (int jj = 0; jj < XJ; ++jj) prep_x(jj);
    };
    // piece j of a stage's fill: j < YJ dY rows, then the X patch
    auto fire_piece = [&](int st, int j) {
        if (WG3_ABL & 4) return;
        if (j < YJ) WG3_BLDS16(yrs, ythr, ysof + (unsigned)j * yjstep, st * STAGE + (YRPI * j + wid * (YRPI / 4)) * YROW);
        else WG3_BLDS16(xrs, xofs[j - YJ], 0, st * STAGE + YBUF + ((j - YJ) * XPPI + wid * (XPPI / 4)) * XPB);
    };
    auto fire = [&](int st) {
#pragma unroll
        for (int j = 0; j < YJ + XJ; ++j) fire_piece(st, j);
    };
    // The K loop is scheduled by hand.  hipcc puts an s_waitcnt vmcnt(0) in front of the first LDS read it can see behind
    // an LDS-DMA (the DMA is a pending LDS write as far as it knows, and it cannot tell the two stages apart): with the
    // fragment reads as compiler-visible loads the "two stages" of gg_wg2.hip were fire -> wait -> multiply, i.e. no
    // overlap at all.  Here the fragment reads and the MFMAs are inline asm, the waits are explicit.
    // (asm volatile statements keep their order; a sched_barrier behind every wait keeps compiler-scheduled code -- the
    //  ReLU of the fragments, the bias sums -- from moving above it, guide rule 18.)
    //
    // The 8-byte halves the reads deliver.  The compiler knows nothing about WHEN an asm read delivers: anything it
    // derives from these registers -- including the copies that assemble two halves into one operand tuple when the
    // allocator did not place them side by side -- must come behind the wait.  So the halves stay as they are until
    // `compose` (called right behind each lgkmcnt(0)) ties them to that point with empty asm statements and only then
    // forms the fragments.  (Composed inside read_frag, an unrelated edit of the tile decode changed the allocation
    // and the bias sums of the <64, 128> form came out wrong, differently from run to run.)
    typedef std::integral_constant<int, 0> K0;
    typedef std::integral_constant<int, 1> K1;
    typedef std::integral_constant<int, 0> St0;
    typedef std::integral_constant<int, 1> St1;
    constexpr bool PINY = PIPE != 0 && WG3_PIN != 0;       // dY fragments in fixed registers (see WG3_PIN)
    bf8_t fa[2][MT], fb[2][NT];
    wg3_u2_t fal[2][MT], fah[2][MT], fbl[2][NT], fbh[2][NT];
    // fragment `i` of k-half KK out of stage ST: i < NT: X tile i, else dY tile i - NT (X first: every MFMA row needs all of
    // them).  The DS offset field is 16 bits: two stages of at most 32 KB + the k-half's 8 KB fit.
    auto read_frag = [&](auto st_tag, auto kk_tag, auto i_tag) {
        constexpr int ST = decltype(st_tag)::value, KK = decltype(kk_tag)::value, i = decltype(i_tag)::value;
        constexpr unsigned SB = (unsigned)(ST * STAGE);
        if (WG3_ABL & 8) return;
        if constexpr (i < NT) {
            unsigned o0 = xbase[KK][0], o1 = xbase[KK][1];
            if (i) { WG3_XOR(o0, xbase[KK][0], i << 5); WG3_XOR(o1, xbase[KK][1], i << 5); }
            wg3_rd_pair<KK * 12 + i, SB, false>(fbl[KK][i], fbh[KK][i], o0, o1);
        } else {
            constexpr int mt = i - NT;
            unsigned a0 = ybase[0], a1 = ybase[1];
            if (mt) { WG3_XOR(a0, ybase[0], mt << 5); WG3_XOR(a1, ybase[1], mt << 5); }
            wg3_rd_pair<KK * 12 + i, SB + KK * (32 * YROW), PINY>(fal[KK][mt], fah[KK][mt], a0, a1);
        }
    };
    auto compose = [&](auto kk_c) {
        constexpr int KK = decltype(kk_c)::value;
        if (WG3_ABL & 8) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) fa[KK][mt] = __builtin_bit_cast(bf8_t, make_uint4(ybase[0], ybase[1], mt, KK));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) fb[KK][nt] = __builtin_bit_cast(bf8_t, make_uint4(xbase[KK][0], xbase[KK][1], nt, KK));
            return;
        }
        if constexpr (!PIPE)     // (the pipelined loop forms the X fragments in its ReLU, one per gap)
            wg3_sfor<NT>([&](auto t) { constexpr int nt = decltype(t)::value; fb[KK][nt] = wg3_compose<KK * 12 + nt, false>(fbl[KK][nt], fbh[KK][nt]); });
        else
            wg3_sfor<NT>([&](auto t) { constexpr int nt = decltype(t)::value; wg3_tie(fbl[KK][nt], fbh[KK][nt]); });
        wg3_sfor<MT>([&](auto t) { constexpr int mt = decltype(t)::value; fa[KK][mt] = wg3_compose<KK * 12 + NT + mt, PINY>(fal[KK][mt], fah[KK][mt]); });
    };
    // bias sums of the dY fragments of k-half kk (a few steps of a few workgroups only)
    auto bias_add = [&](auto kk_tag) {
        constexpr int kk = decltype(kk_tag)::value;
        // wave w adds up the channel tiles w * BT ..: four uniform branches with constant fragment indices (a chain of selects on
        // wid became an indexed copy of the fragments in scratch memory)
        wg3_sfor<4>([&](auto w_tag) {
            constexpr int w = decltype(w_tag)::value;
            if (wid == w) {
                asm volatile("" ::: "memory");       // keeps the four branches apart (merged, they index the fragments again)
#pragma unroll
                for (int i = 0; i < BT; ++i) {
                    const uint4 v = __builtin_bit_cast(uint4, fa[kk][w * BT + i]);
                    const unsigned d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) bsum[i] += __uint_as_float(d[e] << 16) + __uint_as_float(d[e] & 0xffff0000u);
                }
            }
        });
    };
#define WG3_MFMA(mt, nt, kk) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[mt][nt]) : "v"(fa[kk][mt]), "v"(fb[kk][nt]))

    if constexpr (!PIPE) {
        // ---- the round-3 loop (tunable wgrad3_pipe = 0) ---------------------------------------------------------------
        //   step(ST):  reads of k-half 0 -> set 0;  fire the NEXT step's tiles into the other stage;  lgkmcnt(0);
        //              32 MFMAs on set 0 with the reads of k-half 1 -> set 1 between them;  lgkmcnt(0);  32 MFMAs on set 1;
        //              vmcnt(0) (the next tiles have had a whole step to land);  barrier.
        if (kb0 < kb1) {
            prepare(kb0);
            fire(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        auto step = [&](auto st_tag, int kb) {
            constexpr int ST = decltype(st_tag)::value;
            wg3_sfor<NT + MT>([&](auto t) { read_frag(st_tag, K0{}, t); });
            const bool nxt = kb + 1 < kb1;
            if (nxt) {
                prepare(kb + 1);
                fire(ST ^ 1);        // the stage the PREVIOUS step read (behind its barrier)
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            compose(K0{});
            const bool bias_step = do_bias && (kb % jtiles) == jt;
            wg3_sfor<2>([&](auto kk_tag) {
                constexpr int kk = decltype(kk_tag)::value;
                if (xrelu) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) fb[kk][nt] = relu_frag(fb[kk][nt]);
                }
                if (bias_step) bias_add(kk_tag);
                __builtin_amdgcn_sched_barrier(0);
                if (WG3_ABL & 2) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[mt][0][0] += (float)fa[kk][mt][0] + (float)fb[kk][mt % NT][0];
                } else {
                    wg3_sfor<MT * NT>([&](auto t) {
                        constexpr int idx = decltype(t)::value, mt = idx / NT, nt = idx % NT;
                        WG3_MFMA(mt, nt, kk);
                        // the other k-half's fragments, one between two MFMAs (its X tiles first)
                        if constexpr (idx < NT + MT && kk == 0) read_frag(st_tag, K1{}, t);
                    });
                }
                if constexpr (kk == 0) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    compose(K1{});
                }
            });
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the NEXT step's tiles have landed: they have had one step to do so
            __builtin_amdgcn_s_barrier();
        };
        int kb = kb0;
        for (; kb + 1 < kb1; kb += 2) {
            step(St0{}, kb);
            step(St1{}, kb + 1);
        }
        if (kb < kb1) step(St0{}, kb);
    } else {
        // ---- the pipelined loop (round 6, default) ---------------------------------------------------------------------
        // What bounds the loop above with ONE workgroup per CU (the grid beside the input-gradient stream) is the issue rate
        // of a wave that is alone on its SIMD: one instruction per ~4-5 cycles, whatever its kind.  A step carries 64 MFMAs
        // (1024 matrix cycles) and ~280 other instructions (48 fragment reads + their address XORs, the 7 LDS-DMA pieces of
        // the next fill at >= 60 cycles of issue each, the fill's address arithmetic, ReLU, copies, waits): only the k-half-1
        // reads sat between MFMAs, everything else -- and the LDS latency of the k-half-0 reads behind the barrier -- ran with
        // the matrix pipe idle: 2790 cycles per step, 2260 with the fills compiled out (profiles/r06_ab_summary.txt, r06l).
        // Here every instruction of a step has a place in the shadow of an MFMA (16 cycles each, ~3 issue slots):
        //   M0: 32 MFMAs on k-half 0;  between them the reads of k-half 1, the scalar part of the addresses of the fill after
        //       next, the ReLU of the X fragments just read.
        //   vmcnt(0) (this wave's pieces of the NEXT step's tiles, issued one step ago);  barrier: the other stage is
        //       complete and everyone has finished reading this one.
        //   M1: 32 MFMAs on k-half 1;  between them the reads of the NEXT step's k-half 0 (other stage), the pieces of the
        //       fill after next (into this stage) with the in-image test of each X piece one gap ahead, the ReLU.
        // One barrier per step as before, but nothing waits for LDS behind it: fragments are read one k-half ahead.
        //
        // Registers.  128 accumulators + two sets of 12 fragments + their halves did not fit in 256 with the fill's addresses
        // (the first form of this loop spilled the loop's own invariants, and every scratch reload is a vmcnt wait that also
        // waits for the LDS-DMA in flight).  So the fragments along the tile's LONG side (dY for the 128 x 64 wave tile) are
        // not double-buffered: fragment mt is used by NT consecutive MFMAs and its successor is read INTO THE SAME REGISTERS
        // right behind the last of them (an MFMA has read its operands long before a later ds_read delivers).  Only the last
        // one would be read too late for the barrier (everyone must have finished reading the stage): it alternates between
        // two slots, so MT + 1 fixed slots v[220 ..] (WG3_PIN_CASES) hold the long side.  The short side (X: NT fragments used
        // by every row of MFMAs) stays double-buffered and goes through the ReLU -- max(x, rlo) on signed 16-bit lanes with
        // rlo = 0 or -32768 (no ReLU), compiler-generated, so the results sit in the operand tuples without a copy and there
        // is no branch inside the MFMA sequence.
        const short rlo = xrelu ? (short)0 : (short)-32768;
        typedef short s4_t __attribute__((ext_vector_type(4)));
        auto relu_half = [&](wg3_u2_t h) {
            s4_t x = __builtin_bit_cast(s4_t, h);
            const s4_t z = {rlo, rlo, rlo, rlo};
            x = __builtin_elementwise_max(x, z);
            return __builtin_bit_cast(wg3_u2_t, x);
        };
        // X fragment nt of set KK out of its halves (tied to the wait first), through the ReLU
        auto form_b = [&](auto kk_tag, auto nt_tag) {
            constexpr int KK = decltype(kk_tag)::value, nt = decltype(nt_tag)::value;
            if (WG3_ABL & 8) { fb[KK][nt] = __builtin_bit_cast(bf8_t, make_uint4(xbase[KK][0], xbase[KK][1], nt, KK)); return; }
            wg3_tie(fbl[KK][nt], fbh[KK][nt]);
            const wg3_u2_t lo = relu_half(fbl[KK][nt]), hi = relu_half(fbh[KK][nt]);
            fb[KK][nt] = __builtin_bit_cast(bf8_t, make_uint4(lo.x, lo.y, hi.x, hi.y));
        };
        // (every dY tile, every second X tile -- every fourth of the 64 x 128 wave tile's eight: the others keep the XOR, the
        //  loop does not have the registers)
        constexpr bool LX = NT > MT;          // 64 x 128 wave tile: the X fragments are the long side (see there)
        constexpr int XAS = LX ? 4 : 2;
        unsigned yaddr[MT][2], xaddr[2][NT / XAS][2];
        if constexpr (WG3_ADDR) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) yaddr[mt][h] = ybase[h] ^ (unsigned)(mt << 5);
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int nt = 0; nt < NT; nt += XAS) xaddr[kk][nt / XAS][h] = xbase[kk][h] ^ (unsigned)(nt << 5);
            }
        }
        // dY fragment mt of k-half KK lives in slot mt, the last one in slot MT - 1 + KK; slot s = registers v[252 - 4 s ..]
        wg3_u2_t fyl[MT + 1], fyh[MT + 1];
        auto read_y = [&](auto st_tag, auto kk_tag, auto mt_tag) {
            constexpr int ST = decltype(st_tag)::value, KK = decltype(kk_tag)::value, mt = decltype(mt_tag)::value;
            constexpr int slot = mt < MT - 1 ? mt : MT - 1 + KK;
            if (WG3_ABL & 8) { fyl[slot] = (wg3_u2_t){ybase[0], (unsigned)mt}; fyh[slot] = (wg3_u2_t){ybase[1], (unsigned)KK}; return; }
            if constexpr (WG3_ADDR) wg3_rd2_pinned<23 - slot, (unsigned)(ST * STAGE + KK * (32 * YROW))>(fyl[slot], fyh[slot], yaddr[mt][0], yaddr[mt][1]);
            else wg3_rdx_pair_pinned<23 - slot, (unsigned)(ST * STAGE + KK * (32 * YROW))>(fyl[slot], fyh[slot], ybase[0], ybase[1], (unsigned)(mt << 5));
        };
        auto read_x = [&](auto st_tag, auto kk_tag, auto nt_tag) {
            constexpr int ST = decltype(st_tag)::value, KK = decltype(kk_tag)::value, nt = decltype(nt_tag)::value;
            if (WG3_ABL & 8) return;
            if constexpr (WG3_ADDR && (nt % XAS) == 0) wg3_rd2<(unsigned)(ST * STAGE)>(fbl[KK][nt], fbh[KK][nt], xaddr[KK][nt / XAS][0], xaddr[KK][nt / XAS][1]);
            else if constexpr (WG3_ADDR) wg3_rdx_pair<(unsigned)(ST * STAGE)>(fbl[KK][nt], fbh[KK][nt], xaddr[KK][nt / XAS][0], xaddr[KK][nt / XAS][1], (unsigned)((nt % XAS) << 5));
            else wg3_rdx_pair<(unsigned)(ST * STAGE)>(fbl[KK][nt], fbh[KK][nt], xbase[KK][0], xbase[KK][1], (unsigned)(nt << 5));
        };
        // ---- 64 x 128 wave tile (LX): the roles are swapped.  The EIGHT X fragments of a k-half are the long side: MFMA order
        // X tile outermost, each X fragment used by MT = 4 consecutive MFMAs and re-read into its slot behind them (NT + 1
        // pinned slots v[252 - 4 s ..]); the four dY fragments are the double-buffered short side.  The X fragments go through the ReLU IN PLACE (asm on the named registers): fragment 0
        // of the next k-half behind this half's last wait, fragment nt >= 1 two MFMAs in front of its first use.
        wg3_u2_t fxl[NT + 1], fxh[NT + 1];
        const unsigned rlo2 = xrelu ? 0u : 0x80008000u;
        auto read_ys = [&](auto st_tag, auto kk_tag, auto mt_tag) {
            constexpr int ST = decltype(st_tag)::value, KK = decltype(kk_tag)::value, mt = decltype(mt_tag)::value;
            if (WG3_ABL & 8) return;
            wg3_rd2<(unsigned)(ST * STAGE + KK * (32 * YROW))>(fal[KK][mt], fah[KK][mt], yaddr[mt][0], yaddr[mt][1]);
        };
        // (NOT pinned: hipcc copied pinned halves of this loop-carried set out of their registers right behind the reads -- before
        //  the data had arrived -- and back in front of the tie; formed by the compiler behind the tie, a copy per half)
        auto form_ys = [&](auto kk_tag, auto mt_tag) {
            constexpr int KK = decltype(kk_tag)::value, mt = decltype(mt_tag)::value;
            if (WG3_ABL & 8) { fa[KK][mt] = __builtin_bit_cast(bf8_t, make_uint4(ybase[0], ybase[1], mt, KK)); return; }
            fa[KK][mt] = wg3_compose<0, false>(fal[KK][mt], fah[KK][mt]);
        };
        auto read_xl = [&](auto st_tag, auto kk_tag, auto nt_tag) {
            constexpr int ST = decltype(st_tag)::value, KK = decltype(kk_tag)::value, nt = decltype(nt_tag)::value;
            constexpr int slot = nt < NT - 1 ? nt : NT - 1 + KK;
            if (WG3_ABL & 8) { fxl[slot] = (wg3_u2_t){xbase[KK][0], (unsigned)nt}; fxh[slot] = (wg3_u2_t){xbase[KK][1], (unsigned)KK}; return; }
            if constexpr ((nt % XAS) == 0) wg3_rd2_pinned<23 - slot, (unsigned)(ST * STAGE)>(fxl[slot], fxh[slot], xaddr[KK][nt / XAS][0], xaddr[KK][nt / XAS][1]);
            else wg3_rdx_pair_pinned<23 - slot, (unsigned)(ST * STAGE)>(fxl[slot], fxh[slot], xaddr[KK][nt / XAS][0], xaddr[KK][nt / XAS][1], (unsigned)((nt % XAS) << 5));
        };
        auto relu_xl = [&](auto kk_tag, auto nt_tag) {
            constexpr int KK = decltype(kk_tag)::value, nt = decltype(nt_tag)::value;
            constexpr int slot = nt < NT - 1 ? nt : NT - 1 + KK;
            if (WG3_ABL & 8) return;
            wg3_relu_pinned<23 - slot>(fxl[slot], fxh[slot], rlo2);
        };
        constexpr int NP = YJ + XJ;                       // LDS-DMA pieces of a fill (7-8 per wave)
        // the gaps (index of the MFMA in front) of a k-half's 32:
        //   0 .. NT-1              X fragment nt of the next k-half (other set)
        //   NT+1                   the LAST dY fragment of the next k-half (its spare slot)
        //   (mt+1) NT, mt < MT-1   dY fragment mt of the next k-half, right behind the last MFMA on the current one
        //   XW .. XW+NT-1          counted wait for the X reads (the dY reads issued since stay in flight), then one ReLU per gap
        //   30                     lgkmcnt(0): the dY fragments are where the MFMAs of the next k-half name them
        constexpr int XW = 13;                            // (LX: the same plan with the sides swapped)
        constexpr int XW_PENDING = 8;                     // DS instructions issued behind the last short-side read when MFMA XW is issued
        static_assert((NT == 4 && MT == 8) || (NT == 8 && MT == 4), "gap plan");
        constexpr int FIRE0 = NT == 4 ? 14 : 10;          // M1: fill piece j behind MFMA FIRE0 + 2 j (8 pieces end at 28 / 24)
        static_assert(FIRE0 + 2 * (NP - 1) <= 29, "fill pieces fit between the MFMAs of a k-half");
        // one k-half: MFMAs on (fy slots of KK, fb[KK]) with the reads of the following k-half NK out of stage RS between them
        auto half = [&](auto kk_tag, auto rs_tag, bool rd, auto&& extra) {
            constexpr int KK = decltype(kk_tag)::value;
            typedef std::integral_constant<int, KK ^ 1> NK;
            if constexpr (LX) {
                wg3_sfor<MT * NT>([&](auto t) {
                    constexpr int idx = decltype(t)::value, nt = idx / MT, mt = idx % MT;
                    constexpr int slot = nt < NT - 1 ? nt : NT - 1 + KK;
                    if (WG3_ABL & 2) acc[mt][nt][0] += __uint_as_float(fxl[slot].x) + (float)fa[KK][mt][0];
                    else wg3_mfma_pinned_b<23 - slot>(acc[mt][nt], fa[KK][mt], fxl[slot], fxh[slot]);
                    // the ReLU of THIS k-half's X fragment nt >= 1, two MFMAs in front of its first use
                    if constexpr (idx >= 2 && (idx + 2) % MT == 0 && (idx + 2) / MT < NT) {
                        relu_xl(kk_tag, std::integral_constant<int, (idx + 2) / MT>{});
                    }
                    if (rd) {
                        if constexpr (idx < MT) read_ys(rs_tag, NK{}, t);
                        if constexpr (idx == MT + 1) read_xl(rs_tag, NK{}, std::integral_constant<int, NT - 1>{});
                        if constexpr (idx >= MT && idx % MT == 0 && idx / MT - 1 < NT - 1) read_xl(rs_tag, NK{}, std::integral_constant<int, idx / MT - 1>{});
                        if constexpr (idx == XW) {
                            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(XW_PENDING) : "memory");
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        if constexpr (idx >= XW && idx < XW + MT) {
                            form_ys(NK{}, std::integral_constant<int, idx - XW>{});
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        if constexpr (idx == 30) {
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                            __builtin_amdgcn_sched_barrier(0);
                            relu_xl(NK{}, std::integral_constant<int, 0>{});
                        }
                    }
                    extra(t);
                });
                return;
            }
            wg3_sfor<MT * NT>([&](auto t) {
                constexpr int idx = decltype(t)::value, mt = idx / NT, nt = idx % NT;
                constexpr int slot = mt < MT - 1 ? mt : MT - 1 + KK;
                if (WG3_ABL & 2) acc[mt][nt][0] += __uint_as_float(fyl[slot].x) + (float)fb[KK][nt][0];
                else wg3_mfma_pinned<23 - slot>(acc[mt][nt], fyl[slot], fyh[slot], fb[KK][nt]);
                if (rd) {
                    if constexpr (idx < NT) read_x(rs_tag, NK{}, t);
                    if constexpr (idx == NT + 1) read_y(rs_tag, NK{}, std::integral_constant<int, MT - 1>{});
                    if constexpr (idx >= NT && idx % NT == 0 && idx / NT - 1 < MT - 1) read_y(rs_tag, NK{}, std::integral_constant<int, idx / NT - 1>{});
                    if constexpr (idx == XW) {
                        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(XW_PENDING) : "memory");
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if constexpr (idx >= XW && idx < XW + NT) {
                        form_b(NK{}, std::integral_constant<int, idx - XW>{});
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if constexpr (idx == 30) {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                extra(t);
            });
        };
        if (kb0 < kb1) {
            prepare(kb0);
            fire(0);
            if (kb0 + 1 < kb1) {
                prepare(kb0 + 1);
                fire(1);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");       // the first fill has landed, the second is in flight
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            if constexpr (LX) {
                wg3_sfor<MT>([&](auto t) { read_ys(St0{}, K0{}, t); });
                wg3_sfor<NT>([&](auto t) { read_xl(St0{}, K0{}, t); });
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                wg3_sfor<MT>([&](auto t) { form_ys(K0{}, t); });
                relu_xl(K0{}, std::integral_constant<int, 0>{});
            } else {
                wg3_sfor<NT>([&](auto t) { read_x(St0{}, K0{}, t); });
                wg3_sfor<MT>([&](auto t) { read_y(St0{}, K0{}, t); });
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                wg3_sfor<NT>([&](auto t) { form_b(K0{}, t); });
            }
        }
        // hipcc re-loads the kernel arguments the address arithmetic of the loop shifts by right in front of the loop and waits
        // for them at their first use INSIDE it -- an lgkmcnt(0) that also waits for the fragment reads in flight, every step.
        // Used here, they are loaded and waited for here.
        asm volatile("" ::"s"(g.ldh), "s"(g.ldw), "s"(g.lsh), "s"(g.lsw), "s"(g.S), "s"(g.Cout));
        auto pstep = [&](auto st_tag, auto fast_tag, int kb) {
            constexpr int ST = decltype(st_tag)::value;
            constexpr bool FAST = decltype(fast_tag)::value;         // the steady state: both following steps exist
            typedef std::integral_constant<int, ST ^ 1> Other;
            const bool nxt1 = FAST || kb + 1 < kb1, nxt2 = FAST || kb + 2 < kb1;
            __builtin_amdgcn_sched_barrier(0);
            half(K0{}, st_tag, true, [&](auto t) {
                constexpr int idx = decltype(t)::value;
                // the scalar part of the addresses of the fill after next, a few instructions per gap
                if constexpr ((NT == 4 && idx >= 18 && idx < 21) || (NT == 8 && idx >= 25 && idx < 28)) {
                    constexpr int c = idx - (NT == 4 ? 18 : 25);
                    __builtin_amdgcn_sched_barrier(0);
                    if (nxt2) {
                        if constexpr (c == 0) prep_a(kb + 2);
                        else if constexpr (c == 1) prep_b();
                        else prep_c();
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            half(K1{}, Other{}, nxt1, [&](auto t) {
                constexpr int idx = decltype(t)::value;
                if constexpr (idx >= FIRE0 && ((idx - FIRE0) & 1) == 0 && (idx - FIRE0) / 2 < NP) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (nxt2) fire_piece(ST, (idx - FIRE0) / 2);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // (the in-image test of an X piece one gap in front of it: its offset lives for two MFMAs)
                if constexpr (idx + 1 >= FIRE0 + 2 * YJ && ((idx + 1 - FIRE0) & 1) == 0 && (idx + 1 - FIRE0) / 2 < NP) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (nxt2) prep_x((idx + 1 - FIRE0) / 2 - YJ);
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
        };
        typedef std::integral_constant<bool, true> Fast;
        typedef std::integral_constant<bool, false> Tail;
        int kb = kb0;
        for (; kb + 3 < kb1; kb += 2) {
            pstep(St0{}, Fast{}, kb);
            pstep(St1{}, Fast{}, kb + 1);
        }
        if (kb < kb1) pstep(St0{}, Tail{}, kb);
        if (kb + 1 < kb1) pstep(St1{}, Tail{}, kb + 1);
        if (kb + 2 < kb1) pstep(St0{}, Tail{}, kb + 2);
    }
    // MFMA results are read by vector instructions below: the hazard distance is the compiler's job for ITS MFMAs only
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    if (do_bias) {
        // lanes fg = 0..3 of a wave hold the same channel fi over different K rows
#pragma unroll
        for (int i = 0; i < BT; ++i) {
            float t = bsum[i];
            t += __shfl_xor(t, 16, 64);
            t += __shfl_xor(t, 32, 64);
            if (fg == 0) atomicAdd(a.dbias + co0 + (wid * BT + i) * 16 + fi, t);
        }
    }
    // ---- the wave's BMC x CI tile: into this split's slab (plain stores), or into dW ----------------------------
    const int wt = (int)((pg.wt4[ph][q] >> (8 * wid)) & 0xffu);
    float* out = a.slab ? a.slab + (size_t)split * ((size_t)g.Cout * g.wtaps * g.Cin) : a.dw;
    // one wave-uniform switch around the 128 stores (round 6: it sat inside the element loop -- four scalar branches per
    // store, 868 in the kernel); addresses = one 64-bit base per lane + a uniform row step
    float* pw0 = out + (size_t)(co0 + fg * 4) * g.wtaps * g.Cin + (size_t)wt * g.Cin + ci0 + fi;
    const size_t rstep = (size_t)g.wtaps * g.Cin;
    auto store_tile = [&](auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;      // 0 plain store (slab / overwrite), 1 read-modify-write, 2 atomic
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float* prow = pw0 + (size_t)(mt * 16 + r) * rstep;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    float* pw = prow + nt * 16;
                    if (WG3_ABL & 1) { if (acc[mt][nt][r] == 123.456f) *pw = 0.f; }
                    else if (MODE == 0) *pw = acc[mt][nt][r];
                    else if (MODE == 1) *pw += acc[mt][nt][r];
                    else atomicAdd(pw, acc[mt][nt][r]);
                }
            }
    };
    if (a.slab || (splits == 1 && a.overwrite)) store_tile(std::integral_constant<int, 0>{});
    else if (splits == 1) store_tile(std::integral_constant<int, 1>{});
    else store_tile(std::integral_constant<int, 2>{});
}

// dW (+)= sum over the splits' slabs (deterministic: fixed association).  Block = 64 float4 elements x 4 split lanes:
// lane q adds the splits q, q + 4, ... with eight 16-B loads in flight, the four lanes meet in LDS in lane order.  (One
// thread per element with the split loop inside it was a chain of dependent loads: the layers with a small dW and many
// splits -- encoders[1] / D block 1: 0.5 MB x 128 -- ran 128 workgroups for 30 us.)
__global__ __launch_bounds__(256) void wgrad_slab_sum_k(float* __restrict__ dw, const float* __restrict__ slab, int nsplits,
                                                        long n4, long slab_stride4, int overwrite) {
    __shared__ float4 red[3][64];
    const int e = threadIdx.x & 63, q = threadIdx.x >> 6;
    const long i = (long)blockIdx.x * 64 + e;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n4) {
        const float4* s = (const float4*)slab + i;
        int k = q;
        for (; k + 28 < nsplits; k += 32) {
            float4 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = s[(long)(k + 4 * u) * slab_stride4];
#pragma unroll
            for (int u = 0; u < 8; ++u) { v.x += t[u].x; v.y += t[u].y; v.z += t[u].z; v.w += t[u].w; }
        }
        for (; k < nsplits; k += 4) {
            const float4 t = s[(long)k * slab_stride4];
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
    }
    if (q) red[q - 1][e] = v;
    __syncthreads();
    if (q == 0 && i < n4) {
#pragma unroll
        for (int j = 0; j < 3; ++j) { v.x += red[j][e].x; v.y += red[j][e].y; v.z += red[j][e].z; v.w += red[j][e].w; }
        if (!overwrite) {
            const float4 o = ((const float4*)dw)[i];
            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        ((float4*)dw)[i] = v;
    }
}

// The common case of the training step -- 2 (to 4) slabs of 8-32 MB: ONE thread per float4 element and slab set, four elements
// per thread, every load of a thread in flight at once, no LDS, no barrier.  (The kernel above gives such a sum to a quarter
// or half of its threads, one 16-B load each, 3 KB per workgroup behind a barrier: 255 us per step on the weight-gradient
// stream for 654 MB.)  Same association as above: ((s0 + s1) + s2) + s3, then + dW.
template <int NS>
__global__ __launch_bounds__(256) void wgrad_slab_sum_few_k(float* __restrict__ dw, const float* __restrict__ slab, long n4,
                                                            long slab_stride4, int overwrite) {
    constexpr int U = 4;
    const long stride = (long)gridDim.x * 256;
    for (long i0 = (long)blockIdx.x * 256 + threadIdx.x; i0 < n4; i0 += stride * U) {
        float4 t[U][NS], o[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = i0 + u * stride;
#pragma unroll
            for (int k = 0; k < NS; ++k) t[u][k] = i < n4 ? ((const float4*)slab)[(long)k * slab_stride4 + i] : make_float4(0.f, 0.f, 0.f, 0.f);
            if (!overwrite) o[u] = i < n4 ? ((const float4*)dw)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = i0 + u * stride;
            float4 v = t[u][0];
#pragma unroll
            for (int k = 1; k < NS; ++k) { v.x += t[u][k].x; v.y += t[u][k].y; v.z += t[u][k].z; v.w += t[u][k].w; }
            if (!overwrite) { v.x += o[u].x; v.y += o[u].y; v.z += o[u].z; v.w += o[u].w; }
            if (i < n4) ((float4*)dw)[i] = v;
        }
    }
}

int launch_wgrad_slab_sum(float* dw, const float* slab, int nsplits, int64_t n, int overwrite, hipStream_t s) {
    PAI_CHECK((n % 4) == 0, "wgrad slab sum: %lld elements are not a multiple of 4", (long long)n);
    const long n4 = (long)(n / 4);
    if (nsplits >= 2 && nsplits <= 4 && pai_tunable("slab_sum_few", 1)) {
        long blocks = (n4 + 256 * 4 - 1) / (256 * 4);
        if (blocks > 2048) blocks = 2048;
        if (nsplits == 2) PAI_LAUNCH(wgrad_slab_sum_few_k<2>, dim3((unsigned)blocks), dim3(256), 0, s, dw, slab, n4, n4, overwrite);
        else if (nsplits == 3) PAI_LAUNCH(wgrad_slab_sum_few_k<3>, dim3((unsigned)blocks), dim3(256), 0, s, dw, slab, n4, n4, overwrite);
        else PAI_LAUNCH(wgrad_slab_sum_few_k<4>, dim3((unsigned)blocks), dim3(256), 0, s, dw, slab, n4, n4, overwrite);
        PAI_LAUNCH_CHECK();
        return 0;
    }
    PAI_LAUNCH(wgrad_slab_sum_k, dim3((unsigned)((n4 + 63) / 64)), dim3(256), 0, s, dw, slab, nsplits, n4, n4, overwrite);
    PAI_LAUNCH_CHECK();
    return 0;
}

// ---- host side ----------------------------------------------------------------------------------------------
static int wg3_variant(const GG& g) {   // 0: not eligible, 1: <128, 64>, 2: <64, 128>, 3: <128, 64> on 8 x 8-pixel K steps
    const int mode = pai_tunable("wgrad3", 7);   // bit 0: the 128 x 64 wave tile, bit 1: the 64 x 128 one, bit 2: 8 x 8-pixel K steps; 0: round-2 kernels
    if (!mode) return 0;
    PatchGeo pg;
    // 32-bit byte offsets into buffer descriptors: every tensor below 2 GB
    if ((int64_t)g.N * g.H * g.W * (g.C1 > g.C2 ? g.C1 : g.C2) * 2 >= (1ll << 31) || (int64_t)g.N * g.OH * g.OW * g.Cout * 2 >= (1ll << 31))
        return 0;
    if (g.lsw >= 0 && g.lw == 3 && g.lh >= 3 && patch_geo(g, 8, &pg, 8))    // 8-wide images (encoders[4], decoders[3])
        return ((mode & 4) && (g.Cout % 128) == 0 && (g.C1 % 64) == 0 && (g.C2 % 64) == 0 && g.Cin >= 64) ? 3 : 0;
    if (!(g.lsw >= 0 && g.lw >= 4 && g.lh >= 2 && patch_geo(g, 4, &pg))) return 0;
    if ((g.Cout % 128) == 0 && (g.C1 % 64) == 0 && (g.C2 % 64) == 0 && g.Cin >= 64) return (mode & 1) ? 1 : 0;
    if ((g.Cout % 64) == 0 && (g.C1 % 128) == 0 && (g.C2 % 128) == 0 && g.Cin >= 128) return (mode & 2) ? 2 : 0;
    return 0;
}

bool wgrad3_ok(const GG& g) { return wg3_variant(g) != 0; }

struct Wg3Cfg { int bmc, ci, cotiles, jtiles, tiles, psplits, per; };

static Wg3Cfg wg3_cfg(const GG& g, int solo = -1) {      // solo: -1 = as the problem says, 0 / 1 = forced (workspace sizing)
    PatchGeo pg;
    const int v = wg3_variant(g);
    if (v == 3) patch_geo(g, 8, &pg, 8); else patch_geo(g, 4, &pg);
    Wg3Cfg c;
    c.bmc = v == 2 ? 64 : 128;
    c.ci = v == 2 ? 128 : 64;
    c.cotiles = g.Cout / c.bmc;
    c.jtiles = (g.Cin / c.ci) * pg.groups;
    c.tiles = c.cotiles * c.jtiles * g.nphase;
    const int kblocks = g.M / 64;
    // pixel splits: ONE workgroup per CU (256).  Alone on the chip two per CU (512) are faster (scripts/micro/convbench), but
    // in the training step the weight gradients run beside the input-gradient chain of the main stream: with 256
    // workgroups they flush half the slab bytes (32 instead of 64 MB per layer, and wgrad_slab_sum_k reads half) and leave
    // the other half of every CU to the main stream -- same-box step, two interleaved runs each: 6.36 ms at 256, 6.42-6.44
    // at 192, 6.49 at 384, 6.55 at 512, 6.67 at 768, 7.05 at 128.  A split never gets fewer than 512 pixels.
    // PAI_HINT_SOLO (the last weight gradients of a backward pass: the input-gradient chain has ended, the main stream only
    // waits for them): two workgroups per CU, the grid that is fastest alone (D block 1: 292 -> ~180 us in the step's tail).
    if (solo < 0) solo = g.solo;
    int splits = cdiv(solo ? pai_tunable("wgrad3_target_solo", 512) : pai_tunable("wgrad3_target", 256), c.tiles);
    const int max_splits = cdiv(g.M, pai_tunable("wgrad3_minrows", 512));
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    c.per = cdiv(kblocks, splits);
    c.psplits = cdiv(kblocks, c.per);
    return c;
}

int64_t wgrad3_slab_bytes(const GG& g) {
    if (!wgrad3_ok(g)) return 0;
    const Wg3Cfg c0 = wg3_cfg(g, 0), c1 = wg3_cfg(g, 1);     // the workspace serves the layer with and without PAI_HINT_SOLO
    const int ps = c0.psplits > c1.psplits ? c0.psplits : c1.psplits;
    return ps > 1 ? (int64_t)ps * g.Cout * g.wtaps * g.Cin * 4 : 0;
}

// pai_conv_wgrad_overwrite needs no zero fill: every dW element has one writer (un-split: the taps of different phases
// are disjoint) or the slab sum writes it
bool wgrad3_overwrites(const GG& g) {
    if (!wgrad3_ok(g)) return false;
    const Wg3Cfg c = wg3_cfg(g);
    if (c.psplits == 1) return true;
    return pai_tunable("wgrad_slab", 1) && wgrad_slab_acquire((int64_t)c.psplits * g.Cout * g.wtaps * g.Cin * 4) != nullptr;
}

const char* wgrad3_kernel_name(const GG& g) {
    const int v = wg3_variant(g);
    return v == 1 ? "gg_wgrad_patch3_k<128, 64, 16>" : (v == 2 ? "gg_wgrad_patch3_k<64, 128, 16>" : "gg_wgrad_patch3_k<128, 64, 8>");
}

int launch_wgrad3(const GG& g, const WgradArgs& a0, hipStream_t s) {
    PatchGeo pg;
    const int variant = wg3_variant(g);
    PAI_CHECK(variant != 0 && (variant == 3 ? patch_geo(g, 8, &pg, 8) : patch_geo(g, 4, &pg)), "launch_wgrad3: problem not eligible");
    const Wg3Cfg c = wg3_cfg(g);
    WgradArgs a = a0;
    const int64_t dwn = (int64_t)g.Cout * g.wtaps * g.Cin;
    const int64_t need = (int64_t)c.psplits * dwn * 4;
    float* slab = nullptr;
    if (c.psplits > 1 && pai_tunable("wgrad_slab", 1)) slab = wgrad_slab_acquire(need);
    a.slab = slab;
    if (a.overwrite_bias && a.dbias) {   // the bias sums of the workgroups meet by atomics
        hipError_t e = pai::memset_async(a.dbias, 0, (size_t)g.Cout * sizeof(float), s);
        PAI_CHECK(e == hipSuccess, "launch_wgrad3: hipMemsetAsync: %s", hipGetErrorString(e));
    }
    static PerDeviceOnce attr;
    if (attr.first()) {
        hipError_t e = hipSuccess;
        const void* fns[6] = {reinterpret_cast<const void*>(&gg_wgrad_patch3_k<128, 64, 16>), reinterpret_cast<const void*>(&gg_wgrad_patch3_k<64, 128, 16>),
                              reinterpret_cast<const void*>(&gg_wgrad_patch3_k<128, 64, 8>), reinterpret_cast<const void*>(&gg_wgrad_patch3_k<128, 64, 16, 0>),
                              reinterpret_cast<const void*>(&gg_wgrad_patch3_k<64, 128, 16, 0>), reinterpret_cast<const void*>(&gg_wgrad_patch3_k<128, 64, 8, 0>)};
        for (int i = 0; i < 6 && e == hipSuccess; ++i) e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
        PAI_CHECK(e == hipSuccess, "hipFuncSetAttribute(max dynamic LDS): %s", hipGetErrorString(e));
    }
    const dim3 grid(c.tiles * c.psplits);
    const int ph_inner = g.nphase > 1 && pai_tunable("wgrad3_ph_inner", 1);
    // the pipelined K loop (see the kernel): not with a bias gradient (the bias sums live in the round-3 loop only)
    const int pipe = pai_tunable("wgrad3_pipe", 1) && !a.dbias;
    if (variant == 1) {
        const size_t lds = (size_t)2 * (64 * 256 + 96 * 128);
        if (pipe) PAI_LAUNCH((gg_wgrad_patch3_k<128, 64, 16>), grid, dim3(256), lds, s, g, a, pg, c.cotiles, c.jtiles, c.psplits, c.per, ph_inner);
        else PAI_LAUNCH((gg_wgrad_patch3_k<128, 64, 16, 0>), grid, dim3(256), lds, s, g, a, pg, c.cotiles, c.jtiles, c.psplits, c.per, ph_inner);
    } else if (variant == 2) {
        const size_t lds = (size_t)2 * (64 * 128 + 96 * 256);
        if (pipe) PAI_LAUNCH((gg_wgrad_patch3_k<64, 128, 16>), grid, dim3(256), lds, s, g, a, pg, c.cotiles, c.jtiles, c.psplits, c.per, ph_inner);
        else PAI_LAUNCH((gg_wgrad_patch3_k<64, 128, 16, 0>), grid, dim3(256), lds, s, g, a, pg, c.cotiles, c.jtiles, c.psplits, c.per, ph_inner);
    } else {
        const size_t lds = (size_t)2 * (64 * 256 + 128 * 128);
        if (pipe) PAI_LAUNCH((gg_wgrad_patch3_k<128, 64, 8>), grid, dim3(256), lds, s, g, a, pg, c.cotiles, c.jtiles, c.psplits, c.per, ph_inner);
        else PAI_LAUNCH((gg_wgrad_patch3_k<128, 64, 8, 0>), grid, dim3(256), lds, s, g, a, pg, c.cotiles, c.jtiles, c.psplits, c.per, ph_inner);
    }
    PAI_LAUNCH_CHECK();
    if (slab) return launch_wgrad_slab_sum(a.dw, slab, c.psplits, dwn, a.overwrite, s);
    return 0;
}

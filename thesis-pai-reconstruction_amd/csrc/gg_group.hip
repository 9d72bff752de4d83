// Grouped 3x3 "same" convolution of the ResNeXt blocks (nn.Conv2d(128, 128, 3, padding=1, groups=32), reference
// models/res_unet.py:151-157), forward and input gradient, from the BLOCK-DIAGONAL dense filter packs
// (pai_conv_desc.groups).  The dense tile kernels spend 32x the useful MACs on the zero blocks and run the layer
// compute-bound (1.5 ms at 512 x 512 x 16 images); here an output tile of 16 channels only contracts with its own 16
// input channels (K = 9 taps x 16, five MFMA K steps, 4x the useful MACs) and the layer is bound by moving the
// activation once:
//   * a workgroup owns an 8 x 16 block of output pixels x 64 channels (four slices; blockIdx.y picks the half) and
//     keeps the 10 x 18 source pixels x 64 channels it touches in LDS (23 KB, one coalesced fill; 16-B chunk c of
//     patch pixel p at slot c ^ (((p >> 1) & 3) << 1) of its 128-B row: a ds_read_b128 lane group is 8 lanes of one
//     k-quarter (pixels 0-3, 12-15 or 4-11) plus 8 of its neighbour, so bit 0 of the slot -- the k-quarter's
//     channel half -- is kept and the XOR spreads each set of 8 pixels over the 2 x 4 row-parity / slot-pair
//     positions, for every tap shift);
//   * wave w takes slice w over the eight pixel rows of the tile: its filter slice (five 16-B fragments per lane, loaded
//     once per persistent workgroup) is the MFMA A operand, the shifted patch pixels are B, no operand is re-read from
//     memory for the nine taps.
//   * bias, BatchNorm partial statistics (one row per persistent workgroup), activation in the store.
// A first attempt without LDS (per-tap 16-B loads from L1, gg_small.hip in slice mode) ran at the speed of the dense
// kernel: nine shifted re-reads of every pixel line through the texture path cost as much as the wasted MACs.
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf8_t;
typedef __attribute__((ext_vector_type(4))) float f4_t;
typedef __attribute__((ext_vector_type(8))) short s8_t;

#ifndef GROUP_ABL
#define GROUP_ABL 0     // compile-time timing ablations (results WRONG): 1 no output stores, 2 no LDS fragment reads, 4 no patch fetch
#endif
namespace {
constexpr int GC = 128;                 // channels of the tensor
constexpr int HS = 4;                   // 16-channel slices per workgroup (64 channels = 8 chunks = a 128-B patch row)
constexpr int TH = 8, TW = 16;          // output pixels per tile
constexpr int PW = TW + 2, PH = TH + 2; // patch
constexpr int PATCH_PIXELS = PW * PH;   // 180
constexpr int MAX_BLOCKS = 2048;

// prologue (pai_conv_fwd_pro / pai_conv_wgrad_pro): eight channels of one pixel through the producer's BatchNorm + ReLU --
// the same fma -> max -> bf16 rounding as bn_apply_k
__device__ __forceinline__ uint4 pre_chunk(uint4 v, const float* sc, const float* sh, float plo) {
    const unsigned wv[4] = {v.x, v.y, v.z, v.w};
    unsigned o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float lo = fmaxf(fmaf(__uint_as_float(wv[i] << 16), sc[2 * i], sh[2 * i]), plo);
        const float hi = fmaxf(fmaf(__uint_as_float(wv[i] & 0xffff0000u), sc[2 * i + 1], sh[2 * i + 1]), plo);
        o[i] = pk2bf(lo, hi);
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}

__device__ __forceinline__ bf8_t relu8g(bf8_t f) {
    s8_t x = __builtin_bit_cast(s8_t, f);
    const s8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
    return __builtin_bit_cast(bf8_t, __builtin_elementwise_max(x, z));
}
}  // namespace

bool grouped3_ok(int dtype, const GG& g, const FwdArgs& a) {
    if (dtype != PAI_BF16 || g.gslice != 16) return false;
    if (g.nphase != 1 || g.S != 1 || g.OS != 1 || g.ntaps != 9 || g.wtaps != 9) return false;
    if (g.C1 != GC || g.C2 != 0 || g.Cout != GC || g.D2 != 0) return false;
    if ((g.OHg % TH) || (g.OWg % TW) || g.H != g.OHg || g.W != g.OWg) return false;
    if (a.yf32 || a.skip_d1 || a.bz) return false;
    if (a.yact && a.eact != PAI_ACT_NONE && a.eact != PAI_ACT_LRELU && a.eact != PAI_ACT_RELU) return false;
    if (a.pscale && (!a.pshift || g.relu1 || (a.pact != PAI_ACT_NONE && a.pact != PAI_ACT_RELU))) return false;
    return true;
}

static int grouped3_tiles(const GG& g) { return g.N * (g.OHg / TH) * (g.OWg / TW); }

int grouped3_rows(const GG& g) {
    const int t = grouped3_tiles(g);
    return t < MAX_BLOCKS ? t : MAX_BLOCKS;
}

template <bool STATS, bool PRE>
__global__ __launch_bounds__(256, 2) void grouped3_k(GG g, FwdArgs a, int tiles, int tiles_x, int tiles_y) {
    __shared__ __attribute__((aligned(16))) unsigned char patch[PATCH_PIXELS * 128];
    const int c0 = blockIdx.y * (HS * 16);      // first channel of this workgroup
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const bf16_t* x = (const bf16_t*)a.x1;
    const bf16_t* w = (const bf16_t*)a.w;
    bf16_t* yraw = (bf16_t*)a.y1;
    bf16_t* yact = (bf16_t*)a.yact;
    const float eslope = act_slope(a.yact ? a.eact : PAI_ACT_NONE);      // branch-free activation (common.h)

    // this lane's share of a K step: tap 2 ks + (fq >> 1) (tap 9 is padding), 8 channels at (fq & 1) * 8 of the slice
    int pofs[5];
    bool kvalid[5];
    int wtap[5];
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
        const int t = 2 * ks + (fq >> 1);
        kvalid[ks] = t < 9;
        const int tt = kvalid[ks] ? t : 0;
        pofs[ks] = (1 + g.dy[0][tt]) * PW + (1 + g.dx[0][tt]);
        wtap[ks] = g.wt[0][tt];
    }
    // (measured and dropped: partial sums in LDS + a rolled slice loop, 147 VGPRs / three workgroups per CU -- 1361 us
    //  forward, 994 us input gradient at 512 x 512 x 16 against 984 / 916 us for this form at 244 VGPRs / two per CU)
    float csum[4], csq[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) csum[r] = csq[r] = 0.f;
    // wave w = slice w of this workgroup's 64 channels, all eight pixel rows of a tile: its filter slice (five 16-B
    // fragments per lane) and bias are loaded ONCE for the whole (persistent) workgroup.  (Until round 3 every wave walked
    // the four slices over two pixel rows and re-read 20 filter fragments per tile -- a chain of L1 round trips per slice
    // that cost more than the patch fetch and the stores together: 1020 -> ~6xx us at 512 x 512 x 16.)
    const int cs = c0 + 16 * wid;             // first channel of this wave's slice
    bf8_t af[5];
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
        uint4 z = make_uint4(0, 0, 0, 0);
        if (kvalid[ks]) z = *(const uint4*)(w + ((size_t)(cs + fr) * 9 + wtap[ks]) * GC + cs + (fq & 1) * 8);
        af[ks] = __builtin_bit_cast(bf8_t, z);
    }
    float bias4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bias4[r] = a.bias ? a.bias[cs + 4 * fq + r] : 0.f;
    const int chunk = 2 * wid + (fq & 1);

    const int tpi = tiles_x * tiles_y;
    // The patch of the NEXT tile is fetched into registers while this one is multiplied (a workgroup per tile pays the
    // HBM latency of its fill in full otherwise: 9.4 us per tile measured, two workgroups per CU do not hide it).
    constexpr int NF = (PATCH_PIXELS * 8 + 255) / 256;      // 16-B chunks per thread
    uint4 pre[NF];
    auto fetch = [&](int tile) {
        const int n = tile / tpi, rem = tile - n * tpi;
        const int y0 = (rem / tiles_x) * TH, x0 = (rem % tiles_x) * TW;
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            const int i = tid + 256 * j;
            const int p = i >> 3, c = i & 7;
            const int py = p / PW, px = p - py * PW;
            const int iy = y0 - 1 + py, ix = x0 - 1 + px;
            const bool inb = i < PATCH_PIXELS * 8 && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
            const size_t off = inb ? ((size_t)(n * g.H + iy) * g.W + ix) * GC + c0 + c * 8 : 0;
            pre[j] = (GROUP_ABL & 4) ? make_uint4(i, j, 0, 0) : *(const uint4*)(x + off);
            if (!inb) pre[j] = make_uint4(0, 0, 0, 0);
        }
    };
    // prologue: this thread's chunk column c = tid & 7 is the same for every patch chunk it moves
    float psc[PRE ? 8 : 1], psh[PRE ? 8 : 1];
    const float plo = a.pact == PAI_ACT_RELU ? 0.f : -INFINITY;
    if (PRE) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { psc[e] = a.pscale[c0 + (tid & 7) * 8 + e]; psh[e] = a.pshift[c0 + (tid & 7) * 8 + e]; }
    }
    if ((int)blockIdx.x < tiles) fetch(blockIdx.x);
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int n = tile / tpi, rem = tile - n * tpi;
        const int y0 = (rem / tiles_x) * TH, x0 = (rem % tiles_x) * TW;
        // ---- patch: registers -> LDS (180 pixels x 8 chunks, zero outside the image) ---------------
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            const int i = tid + 256 * j;
            const int p = i >> 3, c = i & 7;
            uint4 v = pre[j];
            if (PRE) {          // the producer's BatchNorm + ReLU on the way into LDS; the zero padding stays zero
                const int py = p / PW, px = p - py * PW;
                const bool inb = (unsigned)(y0 - 1 + py) < (unsigned)g.H && (unsigned)(x0 - 1 + px) < (unsigned)g.W;
                if (inb) v = pre_chunk(v, psc, psh, plo);
            }
            if (i < PATCH_PIXELS * 8) *(uint4*)(patch + p * 128 + ((c ^ (((p >> 1) & 3) << 1)) << 4)) = v;
        }
        __syncthreads();
        if (tile + (int)gridDim.x < tiles) fetch(tile + gridDim.x);
        // ---- this wave's slice x the eight pixel rows of the tile ---------------------------------------
#pragma unroll
        for (int pyo = 0; pyo < TH; ++pyo) {
            const int pbase = pyo * PW + fr;
            f4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 5; ++ks) {
                const int p = pbase + pofs[ks];
                uint4 v = (GROUP_ABL & 2) ? make_uint4(p, chunk, ks, pyo)
                                          : *(const uint4*)(patch + p * 128 + ((chunk ^ (((p >> 1) & 3) << 1)) << 4));
                if (!kvalid[ks]) v = make_uint4(0, 0, 0, 0);
                bf8_t b = __builtin_bit_cast(bf8_t, v);
                if (g.relu1) b = relu8g(b);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks], b, acc, 0, 0, 0);
            }
            // D[i = 4 fq + r][j = fr]: channel cs + 4 fq + r of pixel (y0 + pyo, x0 + fr)
            float v4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v4[r] = acc[r] + bias4[r];
            if (STATS) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    csum[r] += v4[r];
                    csq[r] = fmaf(v4[r], v4[r], csq[r]);
                }
            }
            const size_t o = ((size_t)(n * g.OHg + y0 + pyo) * g.OWg + x0 + fr) * GC + cs + 4 * fq;
            if ((GROUP_ABL & 1) && v4[0] != 12345.f) continue;
            if (yraw) *(uint2*)(yraw + o) = make_uint2(pk2bf(v4[0], v4[1]), pk2bf(v4[2], v4[3]));
            if (yact) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v4[r] = act_fwd(v4[r], eslope);
                *(uint2*)(yact + o) = make_uint2(pk2bf(v4[0], v4[1]), pk2bf(v4[2], v4[3]));
            }
        }
        __syncthreads();   // everyone is done with the patch before the next fill
    }
    if (!STATS) return;
    // ---- BatchNorm partial statistics: one row per workgroup (every wave owns its 16 channels) -----------------
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float s1 = csum[r], s2 = csq[r];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            s1 += __shfl_xor(s1, o, 64);
            s2 += __shfl_xor(s2, o, 64);
        }
        if (fr == 0) {
            float* dst = a.stats + ((size_t)blockIdx.x * 2) * GC + cs + 4 * fq + r;
            dst[0] = s1;
            dst[GC] = s2;
        }
    }
}

int launch_grouped3(const GG& g, const FwdArgs& a, hipStream_t s) {
    const int tiles = grouped3_tiles(g);
    const dim3 grid(grouped3_rows(g), GC / (HS * 16));
    if (a.pscale) {
        if (a.stats) PAI_LAUNCH((grouped3_k<true, true>), grid, dim3(256), 0, s, g, a, tiles, g.OWg / TW, g.OHg / TH);
        else PAI_LAUNCH((grouped3_k<false, true>), grid, dim3(256), 0, s, g, a, tiles, g.OWg / TW, g.OHg / TH);
    } else {
        if (a.stats) PAI_LAUNCH((grouped3_k<true, false>), grid, dim3(256), 0, s, g, a, tiles, g.OWg / TW, g.OHg / TH);
        else PAI_LAUNCH((grouped3_k<false, false>), grid, dim3(256), 0, s, g, a, tiles, g.OWg / TW, g.OHg / TH);
    }
    PAI_LAUNCH_CHECK();
    return 0;
}


// ------------------------------------------------------------------------------------------------------------------------
// Weight gradient of the same layer.  Until round 3 it went through the dense gg_wgrad_mfma_k (32x the useful MACs on the
// zero blocks: 1.64 ms at 512 x 512 x 16 images, where reading x and dy once takes 0.36 ms).  Here only the eight diagonal
// 16 x 16 blocks per tap are formed (4x the useful MACs, 1/8 of the dense work):
//   dW[co][tap][ci] = sum_pixels dy[p][co] * x[p + (dy_tap, dx_tap)][ci],   co, ci in one 16-channel slice.
//   * a workgroup owns 4 x 32 output pixels x 64 channels (blockIdx.y picks the half) per step: dy tile (16 KB) and the
//     6 x 34 source pixels it meets (26 KB) in LDS, 128-B rows, 16-B chunk c of row p at slot c ^ gsw(p);
//   * wave w = slice w of the half: one K step = one 32-pixel row of the tile; the dy fragment (A: 16 output channels x 32
//     pixels) and the nine shifted x fragments (B: 32 pixels x 16 input channels) are TRANSPOSED reads of the pixel-major
//     tiles (ds_read_b64_tr_b16, as thin_wgrad_k), nine MFMAs per K step into nine accumulators;
//   * the next tile's 42 KB are fetched into registers while this one is multiplied (as grouped3_k);
//   * persistent workgroups store their 4 x 9 x 16 x 16 partial blocks into the weight-gradient workspace, a second
//     launch sums them in a fixed order (deterministic) and assigns / adds the diagonal blocks of dW.
// Only the diagonal 16-channel blocks of dW are written (the rest of a block-diagonal filter's gradient is not defined;
// the dense kernel used to leave cross-group products there).  No bias gradient (the layer feeds a BatchNorm).
namespace {
constexpr int WTH = 4, WTW = 32;              // output pixels per step
constexpr int WPW = WTW + 2, WPH = WTH + 2;   // source patch 6 x 34
constexpr int WPATCH = WPW * WPH;             // 204 pixels
constexpr int WDY = WTH * WTW;                // 128 pixels
constexpr int WG_MAX_BLOCKS = 512;
constexpr int WPART = 2 * HS * 9 * 256;       // floats per persistent workgroup PAIR (both channel halves)

// 16-B chunk c of row p sits at slot c ^ gsw(p): the four consecutive rows a 16-lane group of a transposed read touches
// land in four different 32-B bank ranges, and the group eight rows further on in the complementary four
__device__ __forceinline__ int gsw(int p) { return ((p & 3) ^ ((p >> 3) & 1)) << 1; }

// fragment "16 channels x 32 rows" of a pixel-major tile (128-B rows): lane (fr = channel, fq = k quarter) ends up with
// rows row0 + 8 fq + 0..7 of channel 16 * (chunk0 / 2) + fr
__device__ __forceinline__ bf8_t tr_frag16(const unsigned char* tile, int row0, int chunk0, int lane) {
    typedef __attribute__((ext_vector_type(4))) __bf16 bf4_t;
    const int fr = lane & 15, fq = lane >> 4, tq = fr >> 2, tp = fr & 3;
    bf8_t f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int row = row0 + fq * 8 + h * 4 + tq;
        const int chunk = chunk0 + (tp >> 1);
        bf4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
            (bf4_t __attribute__((address_space(3)))*)(tile + row * 128 + ((chunk ^ gsw(row)) << 4) + 8 * (tp & 1)));
#pragma unroll
        for (int e = 0; e < 4; ++e) f[h * 4 + e] = v[e];
    }
    return f;
}
}  // namespace

bool grouped3_wgrad_ok(int dtype, const GG& g, const float* dbias) {
    if (dtype != PAI_BF16 || g.gslice != 16 || dbias) return false;
    if (g.nphase != 1 || g.S != 1 || g.OS != 1 || g.ntaps != 9 || g.wtaps != 9) return false;
    if (g.C1 != GC || g.C2 != 0 || g.Cout != GC || g.relu1) return false;
    if ((g.OHg % WTH) || (g.OWg % WTW) || g.H != g.OHg || g.W != g.OWg) return false;
    return pai_tunable("grouped_wgrad", 1) != 0;
}

static int grouped3_wgrad_blocks(const GG& g) {
    const int t = g.N * (g.OHg / WTH) * (g.OWg / WTW);
    return t < WG_MAX_BLOCKS ? t : WG_MAX_BLOCKS;
}

int64_t grouped3_wgrad_part_bytes(const GG& g) { return (int64_t)grouped3_wgrad_blocks(g) * WPART * sizeof(float); }

template <bool PRE>
__global__ __launch_bounds__(256, 2) void grouped3_wgrad_k(GG g, WgradArgs a, float* part, int tiles, int tiles_x, int tiles_y) {
    __shared__ __attribute__((aligned(16))) unsigned char xp[WPATCH * 128];
    __shared__ __attribute__((aligned(16))) unsigned char dyt[WDY * 128];
    const int c0 = blockIdx.y * (HS * 16);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bf16_t* x = (const bf16_t*)a.x1;
    const bf16_t* dy = (const bf16_t*)a.dy;
    int pofs[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) pofs[t] = (1 + g.dy[0][t]) * WPW + 1 + g.dx[0][t];
    f4_t acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = (f4_t){0.f, 0.f, 0.f, 0.f};

    const int tpi = tiles_x * tiles_y;
    constexpr int NX = (WPATCH * 8 + 255) / 256;     // 7
    constexpr int NY = WDY * 8 / 256;                // 4
    uint4 pre[NX + NY];
    auto fetch = [&](int tile) {
        const int n = tile / tpi, rem = tile - n * tpi;
        const int y0 = (rem / tiles_x) * WTH, x0 = (rem % tiles_x) * WTW;
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int i = tid + 256 * j;
            const int p = i >> 3, c = i & 7;
            const int py = p / WPW, px = p - py * WPW;
            const int iy = y0 - 1 + py, ix = x0 - 1 + px;
            const bool inb = i < WPATCH * 8 && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
            const size_t off = inb ? ((size_t)(n * g.H + iy) * g.W + ix) * GC + c0 + c * 8 : 0;
            pre[j] = *(const uint4*)(x + off);
            if (!inb) pre[j] = make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NY; ++j) {
            const int i = tid + 256 * j;
            const int p = i >> 3, c = i & 7;
            const int py = p / WTW, px = p - py * WTW;
            pre[NX + j] = *(const uint4*)(dy + ((size_t)(n * g.H + y0 + py) * g.W + x0 + px) * GC + c0 + c * 8);
        }
    };
    float psc[PRE ? 8 : 1], psh[PRE ? 8 : 1];          // prologue of x, as in grouped3_k
    const float plo = a.pact == PAI_ACT_RELU ? 0.f : -INFINITY;
    if (PRE) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { psc[e] = a.pscale[c0 + (tid & 7) * 8 + e]; psh[e] = a.pshift[c0 + (tid & 7) * 8 + e]; }
    }
    if ((int)blockIdx.x < tiles) fetch(blockIdx.x);
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int rem = tile % tpi;
        const int y0 = (rem / tiles_x) * WTH, x0 = (rem % tiles_x) * WTW;
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int i = tid + 256 * j;
            const int p = i >> 3, c = i & 7;
            uint4 v = pre[j];
            if (PRE) {
                const int py = p / WPW, px = p - py * WPW;
                const bool inb = (unsigned)(y0 - 1 + py) < (unsigned)g.H && (unsigned)(x0 - 1 + px) < (unsigned)g.W;
                if (inb) v = pre_chunk(v, psc, psh, plo);
            }
            if (i < WPATCH * 8) *(uint4*)(xp + p * 128 + ((c ^ gsw(p)) << 4)) = v;
        }
#pragma unroll
        for (int j = 0; j < NY; ++j) {
            const int i = tid + 256 * j;
            const int p = i >> 3, c = i & 7;
            *(uint4*)(dyt + p * 128 + ((c ^ gsw(p)) << 4)) = pre[NX + j];
        }
        __syncthreads();
        if (tile + (int)gridDim.x < tiles) fetch(tile + gridDim.x);
#pragma unroll 1
        for (int y = 0; y < WTH; ++y) {
            const bf8_t af = tr_frag16(dyt, y * WTW, 2 * wid, lane);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const bf8_t bfr = tr_frag16(xp, y * WPW + pofs[t], 2 * wid, lane);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr, acc[t], 0, 0, 0);
            }
        }
        __syncthreads();   // everyone is done with both tiles before the next fill
    }
    // D[i = 4 fq + r][j = fr] = dW[co = cs + i][tap][ci = cs + j], cs = c0 + 16 wid
    const int fr = lane & 15, fq = lane >> 4;
    float* dst = part + (((size_t)blockIdx.x * gridDim.y + blockIdx.y) * HS + wid) * (9 * 256);
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[t * 256 + (4 * fq + r) * 16 + fr] = acc[t][r];
}

struct GroupWt { int wt[9]; };

// dW diagonal blocks = / += sum over the persistent workgroups' partial blocks, in a fixed order: 64 outputs x 4 lanes of
// the workgroup range per block, eight loads in flight per lane
__global__ __launch_bounds__(256) void grouped3_wgrad_reduce_k(float* dw, const float* part, int nblk, GroupWt wt, int overwrite) {
    __shared__ float red[4][64];
    const int e = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
    float sum = 0.f;
    int b = q;
    for (; b + 28 < nblk; b += 32) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = part[(size_t)(b + 4 * u) * WPART + e];
        sum += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    }
    for (; b < nblk; b += 4) sum += part[(size_t)b * WPART + e];
    red[q][threadIdx.x & 63] = sum;
    __syncthreads();
    if (q == 0) {
        sum = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        const int j = e & 15, i = (e >> 4) & 15, t = (e >> 8) % 9, sl = (e >> 8) / 9;     // sl = half * 4 + slice
        const int cs = 16 * sl;
        float* d = dw + ((size_t)(cs + i) * 9 + wt.wt[t]) * GC + cs + j;
        *d = overwrite ? sum : *d + sum;
    }
}

int launch_grouped3_wgrad(const GG& g, const WgradArgs& a, float* part, hipStream_t s) {
    const int tiles = g.N * (g.OHg / WTH) * (g.OWg / WTW);
    const int nblk = grouped3_wgrad_blocks(g);
    if (a.pscale) PAI_LAUNCH(grouped3_wgrad_k<true>, dim3(nblk, GC / (HS * 16)), dim3(256), 0, s, g, a, part, tiles, g.OWg / WTW, g.OHg / WTH);
    else PAI_LAUNCH(grouped3_wgrad_k<false>, dim3(nblk, GC / (HS * 16)), dim3(256), 0, s, g, a, part, tiles, g.OWg / WTW, g.OHg / WTH);
    PAI_LAUNCH_CHECK();
    GroupWt wt;
    for (int t = 0; t < 9; ++t) wt.wt[t] = g.wt[0][t];
    PAI_LAUNCH(grouped3_wgrad_reduce_k, dim3(WPART / 64), dim3(256), 0, s, a.dw, (const float*)part, nblk, wt, a.overwrite);
    PAI_LAUNCH_CHECK();
    return 0;
}

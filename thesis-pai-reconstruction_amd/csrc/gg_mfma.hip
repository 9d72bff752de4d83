// bf16 MFMA gather-GEMM kernels for gfx950 (v_mfma_f32_16x16x32_bf16, fp32 accumulate).  State at the end of round 4;
// which kernel a call takes is decided in launch_fwd_mfma / launch_wgrad_mfma below and reported by pai_conv_kernel_name.
//
// Forward / input-gradient:  out[m][co] = sum_{t,ci} A(m,t,ci) * Wp[co][wt][ci]
//   gg_fwd_patch_k<256|128, 128|64> (+ gg_fwd_patch1_k<256, 64>) -- the kernels of the
//     k4 s2 layers, i.e. of nearly all the FLOPs: a workgroup owns a 16 x 16 (8 x 16) block of output pixels, keeps the
//     source pixels its 2 x 2 tap windows touch in LDS (one patch fill serves four taps) and streams the 64-channel weight
//     tiles through a one- or two-slot ring; LDS-DMA through buffer descriptors, fragments by conflict-free ds_read_b128,
//     eight waves of 64 x 64; epilogue staged through LDS: bias, BatchNorm partial statistics, activation, the fused
//     backward of the producing layer (pai_conv_dgrad_act / _bn).  18 launches per Pix2Pix step, 0.32-0.33 of the nominal
//     bf16 peak in the step, 0.40-0.41 alone (DESIGN.md sections 9-11 for what bounds it).
//   gg_fwd_mfma_k<128, 128|64, SPLITK, DB> -- the tile kernel everything else falls to (1 x 1 / 3 x 3 layers, nn.Linear as a
//     one-tap convolution, shapes without a patch geometry): 128 x BN x 64 tile, both operand tiles HBM -> LDS directly
//     (global_load_lds_dwordx4; out-of-image taps fetch a zero line), LDS image XOR-swizzled through the source address.
//     Layers with few output tiles and a long reduction (the U-Net bottleneck, the ViT projections) are split over K into
//     fp32 slabs of the handle's workspace and finished by splitk_finish_k (slab sum in split order + the same epilogue).
//   pw_k<64, 32> / <32, 64> -- the finest attention gate's pointwise convolutions: no LDS, the filter in registers.
// Weight gradient:           dW[co][wt][ci] (+)= sum_m dY[m][co] * A(m,t,ci)
//   gg_wgrad_patch3_k (gg_wg3.hip) takes the k4 s2 layers; here: gg_wgrad_patch_k (its predecessor, tunable wgrad3 = 0)
//   and gg_wgrad_mfma_k<128|64> for the rest -- both operands are pixel-major in HBM (NHWC), i.e. K-strided: staged
//   row-major into LDS and fetched as MFMA fragments with ds_read_b64_tr_b16; pixel splits meet by fp32 atomics
//   (un-split launches store).
//
// Serves the dense layers of the reference's hot path: EncoderBlock / DecoderBlock convs
// (models/pix2pix.py:58-111), DiscriminatorBlock 1-3 (models/wrapper.py:229-232), the 1 x 1 / 3 x 3 convolutions and
// nn.Linear layers of the other families (models/attention_unet.py:72-84, models/res_unet.py:147-163,
// models/trans_unet.py:143-156) and their aten::convolution_backward calls.
#include <stdlib.h>

#include <type_traits>

#include "gg_tile.h"


bool fwd_mfma_ok(int dtype, const GG& g, const FwdArgs& a) {
    if (dtype != PAI_BF16) return false;
    if (g.C1 % 64 || g.C2 % 64) return false;
    if (g.Cout % 64) return false;
    if (g.D2 > 0 && (g.D1 % 64)) return false;
    if (a.yf32) return false;
    if (a.skip_d1) return false;
    if ((a.y1 || a.y2) && a.yact) return false;  // one storage-dtype output per launch
    if (a.yact && a.eact != PAI_ACT_NONE && a.eact != PAI_ACT_LRELU && a.eact != PAI_ACT_RELU) return false;
    return true;
}

// Tile configuration of one launch.  L2 -> LDS fill bandwidth (~70 GB/s per CU) is what bounds this
// kernel, so the tile is made as large as the problem allows: 256 x 128 (85 FLOP per staged byte,
// double-buffered, one 512-thread workgroup per CU) when that still gives every CU a workgroup,
// else 128 x 128 / 128 x 64 (single buffer, 3-4 workgroups per CU), split over K when even that
// leaves CUs idle.
// (A 128 x 64 wave-tile form of the 256-row kernel, gg_fwd_patchw_k / tunable fwd_wide, lived here in rounds 3-4: 5-9 %
// faster when a launch has the GPU to itself, 1 % slower over the step -- beside the weight-gradient stream a CU holds one
// forward workgroup, and eight 64 x 64 waves hide more latency than four 128 x 64 ones.  Removed in round 5; the body
// still takes WPX = 128.)
struct FwdCfg { int bm, bn, ksplit; };

static FwdCfg fwd_cfg(const GG& g) {
    FwdCfg c;
    c.bn = ((g.Cout % 128) == 0 && (g.D2 == 0 || (g.D1 % 128) == 0)) ? 128 : 64;
    c.bm = 128;
    c.ksplit = 1;
    const int ntiles = g.Cout / c.bn;
    const int niter = g.ntaps * g.Cin / MBK;
    static const int mode = getenv("PAI_FWD_MODE") ? atoi(getenv("PAI_FWD_MODE")) : 0;
    if (mode == 1 && c.bn == 128 && (int64_t)cdiv(g.M, 256) * ntiles * g.nphase >= 256 && niter >= 4) {
        c.bm = 256;
        return c;
    }
    if (mode == 2 && c.bn == 128 && (int64_t)cdiv(g.M, 128) * ntiles * g.nphase >= 512 && niter >= 4) {
        c.bm = -128;   // 128 x 128, double-buffered
        return c;
    }
    const int tiles = cdiv(g.M, 128) * ntiles * g.nphase;
    if (tiles >= 768 || niter < 8) return c;
    static const int fixed_env = getenv("PAI_FWD_KSPLIT") ? atoi(getenv("PAI_FWD_KSPLIT")) : 0;
    const int fixed = pai_tunable("fwd_ksplit", fixed_env);
    if (fixed > 0) {
        int ks = fixed > niter / 2 ? niter / 2 : fixed;
        while (ks > 1 && (niter % ks)) --ks;
        c.ksplit = ks;
        return c;
    }
    // Split count by a small cost model (us), fitted to the per-layer timings of scripts/bench_conv.py:
    // a K iteration of one workgroup takes ~1.0 us alone on its CU and ~1.3 us with three resident
    // (which then progress together); a split adds one fp32 slab written and read back at ~5 TB/s plus
    // the finish launch.  Every split gets the same number of iterations (ks divides niter).
    const double out_bytes = (double)g.nphase * g.M * g.Cout * 4.0;
    double best = 1e30;
    int best_ks = 1;
    for (int ks = 1; ks <= niter / 2; ++ks) {
        if (niter % ks) continue;
        const double wgs = (double)tiles * ks;
        double t;
        if (wgs <= 768.0) {
            const double occ = wgs <= 256.0 ? 1.0 : wgs / 256.0;
            t = (niter / ks) * (0.85 + 0.15 * occ);
        } else {
            t = cdiv((int64_t)wgs, 768) * (niter / ks) * 1.3;
        }
        if (ks > 1) t += 3.0 + 2.0 * ks * out_bytes / 5e6;
        if (t < best) { best = t; best_ks = ks; }
    }
    c.ksplit = best_ks;
    return c;
}

constexpr int FIN_ROWS = 16;   // rows per split-K finish workgroup (= granularity of its BN partial statistics)

int fwd_mfma_ksplit(const GG& g) { return fwd_cfg(g).ksplit; }

int64_t fwd_mfma_workspace_bytes(const GG& g) {
    const int ks = fwd_cfg(g).ksplit;
    if (ks <= 1) return 0;
    return (int64_t)ks * g.nphase * g.M * g.Cout * 4;   // one fp32 slab per K split
}

// the split actually used: only when the registered scratch is large enough
static int fwd_effective_ksplit(const GG& g) {
    const int ks = fwd_cfg(g).ksplit;
    if (ks > 1 && (pai_ctx()->workspace == nullptr || pai_ctx()->workspace_bytes < fwd_mfma_workspace_bytes(g))) return 1;
    return ks;
}

int fwd_mfma_ksplit_effective(const GG& g) { return fwd_effective_ksplit(g); }

// number of BN partial-statistics rows per phase this launch configuration writes
// rows per workgroup of the patch-resident kernel for this problem: 0 (not applicable), 128 or 256
static int patch_rows(const GG& g, const FwdCfg& c);

int fwd_mfma_mtiles(const GG& g) {
    if (fwd_effective_ksplit(g) > 1) return cdiv(g.M, FIN_ROWS);
    const FwdCfg c = fwd_cfg(g);
    return patch_rows(g, c) == 256 ? g.M / 256 : cdiv(g.M, abs(c.bm));
}


// BM x BN x 64 tile, BM/64 x 2 waves of 64 x (BN/2).  DB = double-buffered LDS: the LDS-DMA of tile
// k+1 is issued before tile k is consumed and retired with a COUNTED s_waitcnt vmcnt + raw s_barrier
// (a __syncthreads() would drain it: guide "Pipelining across barriers").

#ifndef FWDK_SETPRIO
#define FWDK_SETPRIO 0   // 1: s_setprio 1 around the MFMA clusters of gg_fwd_mfma_k (as PATCH_SETPRIO); measured in the step: 5.697-5.701 against 5.664-5.680 ms -- not used
#endif
template <int BM, int BN, bool SPLITK, bool DB, int WR = 64>   // WR: output rows per wave (64 or 32)
__global__ __launch_bounds__(BM / WR * 128) void gg_fwd_mfma_k(GG g, FwdArgs a, int mtiles, int ntiles, int ksplit,
                                                               float* ws) {
    constexpr int NTHR = BM / WR * 128;
    constexpr int MT = WR / 16;              // 16-row MFMA tiles per wave along M
    constexpr int NT = BN / 32;              // 16-col MFMA tiles per wave along N
    constexpr int RPP = NTHR / 8;            // tile rows covered by one load instruction of the block
    constexpr int AJ = BM / RPP;             // = 4 load instructions per thread for the A tile
    constexpr int BJ = BN / RPP;             // load instructions per thread for the B tile
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    // tile order: column tile fastest, then the 4 output phases of the same source rows (they read
    // the same input pixels), then the row tile -> neighbours in this order share A rows in L2
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int bn = bid % ntiles;
    bid /= ntiles;
    const int ph = bid % g.nphase;
    bid /= g.nphase;
    const int bm = bid % mtiles;
    const int ks = bid / mtiles;
    const int m0 = bm * BM, n0 = bn * BN;

    const bf16_t* x1 = (const bf16_t*)a.x1;
    const bf16_t* x2 = (const bf16_t*)a.x2;
    const bf16_t* w = (const bf16_t*)a.w;
    const bf16_t* zero = (const bf16_t*)g_zero_line;

    // ---- load map: lane-linear LDS image; position (row sr+RPP*j, slot sc) holds global chunk sc^swz
    // The steady state of the K loop is kept almost free of vector-ALU work (it competes with the
    // MFMA issue slots): per tile row only a 32-bit element offset and two 5-bit validity masks are
    // kept; pointers are rebuilt once per (tap, source tensor) segment and then just advanced by
    // 64 channels per K-step.
    const int sc = lane & 7, sr = wid * 8 + (lane >> 3);
    const int gch = (sc ^ ((sr >> 1) & 7)) * 8;  // element offset of the chunk this lane fetches
    int pofs1[AJ], pofs2[AJ];
    unsigned vym[AJ], vxm[AJ];  // bit (d+2): source row / column (grid*S + d) is inside the image
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        const int m = m0 + sr + RPP * j;
        int n, gy, gx;
        decode_row(g, m < g.M ? m : 0, n, gy, gx);
        const int pix = (n * g.H + gy * g.S) * g.W + gx * g.S;
        pofs1[j] = pix * g.C1 + gch;
        pofs2[j] = pix * g.C2 + gch;
        unsigned my = 0, mx = 0;
#pragma unroll
        for (int dd = -2; dd <= 2; ++dd) {
            if (m < g.M && (unsigned)(gy * g.S + dd) < (unsigned)g.H) my |= 1u << (dd + 2);
            if ((unsigned)(gx * g.S + dd) < (unsigned)g.W) mx |= 1u << (dd + 2);
        }
        vym[j] = my;
        vxm[j] = mx;
    }
    const bf16_t* wrow[BJ];
#pragma unroll
    for (int j = 0; j < BJ; ++j) wrow[j] = w + (size_t)(n0 + sr + RPP * j) * g.wtaps * g.Cin + gch;

    const int cchunks = g.Cin / MBK;
    const int niter = g.ntaps * cchunks / ksplit;
    int t = (ks * niter) / cchunks;
    int c0 = ((ks * niter) - t * cchunks) * MBK;

    // ---- fragment read addresses ----------------------------------------------------------------
    const int fr = lane & 15, fq = lane >> 4;
    const int fswz = fr >> 1;
    const unsigned a_base = (unsigned)((wm * WR + fr) * 128);
    const unsigned b_base = (unsigned)(A_BYTES + (wn * (BN / 2) + fr) * 128);

    f4_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f4_t){0.f, 0.f, 0.f, 0.f};

    int relu_cur = 0;
    const bf16_t* pa[AJ];
    int astep[AJ];
    const bf16_t* pb[BJ];
    int seg_left = 0, seg_relu = 0;
    // prepare(): (re)build the row pointers when the next tile starts a new (tap, source) segment.
    // It is called right AFTER the loads of the current tile have been fired, so its scalar loads and
    // vector-ALU work run while the LDS-DMA is in flight instead of delaying the next fire().
    auto prepare = [&]() {
        if (seg_left != 0) return;  // wave-uniform
        const int ddy = g.dy[ph][t], ddx = g.dx[ph][t];
        const int dpix = ddy * g.W + ddx;
        const unsigned sy = (unsigned)(ddy + 2), sx = (unsigned)(ddx + 2);
        if (c0 < g.C1) {
            const int sofs = dpix * g.C1 + c0;
            seg_left = (g.C1 - c0) / MBK;
            seg_relu = g.relu1;
#pragma unroll
            for (int j = 0; j < AJ; ++j) {
                const bool v = ((vym[j] >> sy) & (vxm[j] >> sx) & 1u) != 0;
                pa[j] = v ? x1 + (pofs1[j] + sofs) : zero;
                astep[j] = v ? MBK : 0;
            }
        } else {
            const int sofs = dpix * g.C2 + (c0 - g.C1);
            seg_left = (g.Cin - c0) / MBK;
            seg_relu = g.relu2;
#pragma unroll
            for (int j = 0; j < AJ; ++j) {
                const bool v = ((vym[j] >> sy) & (vxm[j] >> sx) & 1u) != 0;
                pa[j] = v ? x2 + (pofs2[j] + sofs) : zero;
                astep[j] = v ? MBK : 0;
            }
        }
        const int woff = g.wt[ph][t] * g.Cin + c0;
#pragma unroll
        for (int j = 0; j < BJ; ++j) pb[j] = wrow[j] + woff;
    };
    // fire(): issue the LDS-DMA of the prepared tile into stage `buf`, advance to the next tile
    auto fire = [&](int buf) -> int {
        unsigned char* As = smem + buf * STAGE;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            GLDS16(pa[j], As + (j * RPP + wid * 8) * 128);
            pa[j] += astep[j];
        }
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            GLDS16(pb[j], As + A_BYTES + (j * RPP + wid * 8) * 128);
            pb[j] += MBK;
        }
        const int relu = seg_relu;
        --seg_left;
        c0 += MBK;
        if (c0 == g.Cin) { c0 = 0; ++t; }
        return relu;
    };
    auto compute = [&](int buf, int relu) {
        const unsigned char* St = smem + buf * STAGE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const unsigned coff = (unsigned)(((kk * 4 + fq) ^ fswz) << 4);
            bf8_t af[MT], bfr[NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) af[mt] = *(const bf8_t*)(St + a_base + mt * 16 * 128 + coff);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bfr[nt] = *(const bf8_t*)(St + b_base + nt * 16 * 128 + coff);
            if (relu) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) af[mt] = relu_frag(af[mt]);
            }
            if (FWDK_SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt], bfr[nt], acc[mt][nt], 0, 0, 0);
            if (FWDK_SETPRIO) __builtin_amdgcn_s_setprio(0);
        }
    };

    prepare();
    if (DB) {
        relu_cur = fire(0);
        for (int it = 0; it < niter; ++it) {
            int relu_next = 0;
            if (it + 1 < niter) {
                prepare();
                relu_next = fire((it + 1) & 1);
                // all but the AJ+BJ loads just issued have landed -> tile `it` is complete
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AJ + BJ) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            compute(it & 1, relu_cur);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // stage (it & 1) may be overwritten from the next iteration on
            relu_cur = relu_next;
        }
    } else {
        for (int it = 0; it < niter; ++it) {
            relu_cur = fire(0);
            if (it + 1 < niter) prepare();  // overlaps the DMA latency
            __syncthreads();  // drains the LDS-DMA (vmcnt(0)) and publishes the tile
            compute(0, relu_cur);
            __syncthreads();  // every wave is done reading before the next tile overwrites the buffer
        }
    }

    if (SPLITK) {
        // fp32 partial tile -> this split's own slab [split][phase][m][Cout], plain stores: float atomics
        // run at ~1.3 TB/s chip-wide against ~6 TB/s for stores, and at 16-64 splits the added bytes
        // (splits x output) were the whole cost of the bottleneck layers.  splitk_finish_k sums the slabs
        // in a fixed order (deterministic) and applies the epilogue.
        float* dst = ws + ((size_t)(ks * g.nphase + ph) * g.M + m0) * g.Cout + n0;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wm * WR + mt * 16 + fq * 4 + r;
                if (m0 + row < g.M) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        dst[(size_t)row * g.Cout + wn * (BN / 2) + nt * 16 + fr] = acc[mt][nt][r];
                }
            }
        return;
    }

    // ---- epilogue: bias, BN partial statistics, activation, LDS-staged row stores --------------
    constexpr int CROW = BN * 2 + 16;  // padded bytes per staged output row
    constexpr int WM = BM / WR;
    unsigned char* Cs = smem;
    float* sstat = (float*)(smem + BM * CROW);  // [WM][2][BN]
    const float eslope = act_slope(a.yact ? a.eact : PAI_ACT_NONE);
    float csum[NT], csq[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int col = wn * (BN / 2) + nt * 16 + fr;
        const float b = a.bias ? a.bias[n0 + col] : 0.f;
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wm * WR + mt * 16 + fq * 4 + r;
                float v = acc[mt][nt][r] + b;
                if (m0 + row < g.M) { s += v; q += v * v; }
                v = act_fwd(v, eslope);   // branch-free (gg_tile.h): slope 1 = no activation
                *(bf16_t*)(Cs + row * CROW + col * 2) = f2bf(v);
            }
        }
        csum[nt] = s;
        csq[nt] = q;
    }
    if (a.stats) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float s = csum[nt], q = csq[nt];
            s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
            q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
            if (fq == 0) {
                const int col = wn * (BN / 2) + nt * 16 + fr;
                sstat[(wm * 2 + 0) * BN + col] = s;
                sstat[(wm * 2 + 1) * BN + col] = q;
            }
        }
    }
    __syncthreads();
    if (a.stats && tid < BN) {
        float* dst = a.stats + ((size_t)(ph * mtiles + bm) * 2) * g.Cout + n0 + tid;
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int i = 0; i < WM; ++i) { s += sstat[(i * 2 + 0) * BN + tid]; q += sstat[(i * 2 + 1) * BN + tid]; }
        dst[0] = s;
        dst[g.Cout] = q;
    }
    bf16_t* dst;
    int dstride, dcol;
    if (a.yact) { dst = (bf16_t*)a.yact; dstride = g.Cout; dcol = n0; }
    else if (n0 < g.D1) { dst = (bf16_t*)a.y1; dstride = g.D1; dcol = n0; }
    else { dst = (bf16_t*)a.y2; dstride = g.D2; dcol = n0 - g.D1; }
    // backward of the producer fused into the store: same values as storing bf16 and running
    // pai_act_bwd / pai_bn_bwd_reduce over it (the product is formed from the bf16-rounded gradient)
    const bool bwd = a.bz && !a.yact && n0 < g.D1;   // uniform per workgroup
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
    // The chunks of the producer's tensors are requested four passes at a time, ahead of that batch's
    // stores: the stores may alias them as far as the compiler knows, and one load -> store round trip per
    // pass serialised 8 HBM latencies at the tail of every workgroup.
    constexpr int NP = BM / ORP, NB = NP < 4 ? NP : 4;
    if (bwd) bwd_load_params(a, dcol + oc * 8, BP);
#pragma unroll
    for (int p0 = 0; p0 < NP; p0 += NB) {
        size_t offs[NB];
        uint4 zq[NB], aq[NB];
#pragma unroll
        for (int p = 0; p < NB; ++p) {
            const int m = m0 + orow0 + (p0 + p) * ORP;
            int n, gy, gx;
            decode_row(g, m < g.M ? m : 0, n, gy, gx);
            const size_t pix = (size_t)(n * g.OH + gy * g.OS + g.poy[ph]) * g.OW + gx * g.OS + g.pox[ph];
            offs[p] = pix * dstride + dcol + oc * 8;
            if (bwd) {
                zq[p] = *(const uint4*)(bzp + offs[p]);
                aq[p] = bap ? *(const uint4*)(bap + offs[p]) : make_uint4(0, 0, 0, 0);
            }
        }
#pragma unroll
        for (int p = 0; p < NB; ++p) {
            const int row = orow0 + (p0 + p) * ORP;
            if (m0 + row < g.M) {
                uint4 o = *(const uint4*)(Cs + row * CROW + oc * 16);
                if (bwd)
                    o = bwd_chunk(o, zq[p], aq[p], bap != nullptr, a.bscale != nullptr, bsum, a.bact1, a.bact2, BP, bs1, bs2);
                *(uint4*)(dst + offs[p]) = o;
            }
        }
    }
    if (bsum)
        bwd_write_partials<BN, CPR, NTHR / 64>(sstat, bs1, bs2, tid,
                                               a.bpart + ((size_t)(ph * mtiles + bm) * 2) * g.D1 + n0, g.D1,
                                               a.bmean + n0, a.brstd + n0);
}

// Split-K epilogue: sums the fp32 slabs [split][phase][M][Cout] in split order, then bias, BN partial
// statistics per 16-row tile, activation, bf16 store (two destinations, optional activation-backward
// mask).  block = 16 rows x 128 channels, thread = (row, 8-channel group): the `ksplit` 32-B loads of a
// thread are independent, statistics meet through LDS (no atomics, deterministic).
constexpr int FIN_COLS = 128;
__global__ __launch_bounds__(256) void splitk_finish_k(GG g, FwdArgs a, const float* ws, int mtiles, int ksplit) {
    __shared__ float red[2][FIN_ROWS][FIN_COLS + 8];
    const int tid = threadIdx.x;
    const int bm = blockIdx.x, ph = blockIdx.y, cb = blockIdx.z;
    const int row = tid >> 4, cgl = tid & 15;
    const int m = bm * FIN_ROWS + row;
    const int c0 = cb * FIN_COLS + cgl * 8;
    const bool valid = m < g.M && c0 < g.Cout;
    const float eslope = act_slope(a.yact ? a.eact : PAI_ACT_NONE);
    float v[8], bs1[8], bs2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = bs1[k] = bs2[k] = 0.f;
    if (valid) {
        const size_t slab = (size_t)g.nphase * g.M * g.Cout;
        const float* src = ws + ((size_t)ph * g.M + m) * g.Cout + c0;
#pragma unroll 4
        for (int s = 0; s < ksplit; ++s) {
            const float4 v0 = *(const float4*)(src + s * slab), v1 = *(const float4*)(src + s * slab + 4);
            v[0] += v0.x; v[1] += v0.y; v[2] += v0.z; v[3] += v0.w;
            v[4] += v1.x; v[5] += v1.y; v[6] += v1.z; v[7] += v1.w;
        }
        if (a.bias) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] += a.bias[c0 + k];
        }
    }
    if (a.stats) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            red[0][row][cgl * 8 + k] = v[k];          // rows beyond M contribute 0
            red[1][row][cgl * 8 + k] = v[k] * v[k];
        }
    }
    if (valid) {
        unsigned packed[4];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = act_fwd(v[k], eslope);
#pragma unroll
        for (int k = 0; k < 4; ++k) packed[k] = pk2bf(v[2 * k], v[2 * k + 1]);
        int n, gy, gx;
        decode_row(g, m, n, gy, gx);
        const size_t pix = (size_t)(n * g.OH + gy * g.OS + g.poy[ph]) * g.OW + gx * g.OS + g.pox[ph];
        bf16_t* dst;
        if (a.yact) dst = (bf16_t*)a.yact + pix * g.Cout + c0;
        else if (c0 < g.D1) dst = (bf16_t*)a.y1 + pix * g.D1 + c0;
        else dst = (bf16_t*)a.y2 + pix * g.D2 + (c0 - g.D1);
        uint4 o = make_uint4(packed[0], packed[1], packed[2], packed[3]);
        if (a.bz && !a.yact && c0 < g.D1) {
            BwdParams BP;
            bwd_load_params(a, c0, BP);
            const size_t off = pix * g.D1 + c0;
            const bf16_t* bap = (const bf16_t*)a.badd;
            o = bwd_chunk(o, *(const uint4*)((const bf16_t*)a.bz + off),
                          bap ? *(const uint4*)(bap + off) : make_uint4(0, 0, 0, 0), bap != nullptr,
                          a.bscale != nullptr, a.bpart != nullptr, a.bact1, a.bact2, BP, bs1, bs2);
        }
        *(uint4*)dst = o;
    }
    if (a.stats) {
        __syncthreads();
        const int c = cb * FIN_COLS + tid;
        if (tid < FIN_COLS && c < g.Cout) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int r = 0; r < FIN_ROWS; ++r) { t1 += red[0][r][tid]; t2 += red[1][r][tid]; }
            float* dst = a.stats + ((size_t)(ph * mtiles + bm) * 2) * g.Cout + c;
            dst[0] = t1;
            dst[g.Cout] = t2;
        }
    } else if (a.bpart && a.bz) {
        // BatchNorm-backward partial sums of the D1 channels, one row per 16-row tile (rows beyond M and
        // channels of the D2 part contribute zeros and are not written)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            red[0][row][cgl * 8 + k] = bs1[k];
            red[1][row][cgl * 8 + k] = bs2[k];
        }
        __syncthreads();
        const int c = cb * FIN_COLS + tid;
        if (tid < FIN_COLS && c < g.D1) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int r = 0; r < FIN_ROWS; ++r) { t1 += red[0][r][tid]; t2 += red[1][r][tid]; }
            float* dst = a.bpart + ((size_t)(ph * mtiles + bm) * 2) * g.D1 + c;
            dst[0] = t1;
            dst[g.D1] = a.brstd[c] * (t2 - a.bmean[c] * t1);
        }
    }
}


// ------------------------------------------------------------------------------------
// Patch-resident forward / input-gradient kernel
// ------------------------------------------------------------------------------------
// Measured on the kernel above (profiles/README.md, "LDS budget"): the LDS port, not the matrix
// pipe, bounds it.  Per 128 x 128 x 64 step a workgroup needs 512 MFMA cycles per SIMD but asks
// the LDS for 32 KB of LDS-DMA fill (64 B/clk/CU measured -> 512 cycles) plus 64 KB of fragment
// reads (256 B/clk -> 256 cycles).  The fill is the expensive half, and three quarters of the A
// fill is redundant: the 4 taps of a 2 x 2 window read the same source pixels shifted by one.
// Here a workgroup owns an 8 x 16 block of output pixels of one image and keeps the 9 x 17 source
// pixels its 2 x 2 window touches (64 channels) in LDS: one 20 KB patch fill serves 4 taps, only
// the 16 KB weight tile changes per tap (21 KB of fill per step instead of 32 KB).
//   ConvTranspose2d forward and Conv2d input gradient: each output phase has exactly one 2 x 2 window.
//   Conv2d k4 s2 forward and ConvTranspose2d input gradient: the 16 taps are 4 windows, one per
//   parity plane of the source (patch pixel stride 2).
// LDS image of the patch: pixel p = py * 17 + px at byte 128 p, 16-B chunk c of the pixel stored at
// slot c ^ (p & 6).  With that swizzle the 16 lanes of a ds_read_b128 group -- 16 consecutive p,
// starting anywhere -- hit 16 different bank quads for every tap shift (the XOR only touches chunk
// bits 1-2, the lane's own k-quarter keeps bit 0).

static int patch_rows(const GG& g, const FwdCfg& c) {
    static const bool no_patch = getenv("PAI_NO_PATCH") && atoi(getenv("PAI_NO_PATCH")) != 0;
    static const bool no_256 = getenv("PAI_NO_PATCH256") && atoi(getenv("PAI_NO_PATCH256")) != 0;
    if (no_patch || c.ksplit > 1 || c.bm != 128) return 0;
    // the kernel addresses its sources with 32-bit byte offsets into buffer descriptors
    if ((int64_t)g.N * g.H * g.W * (g.C1 > g.C2 ? g.C1 : g.C2) * 2 >= (1ll << 31) || (int64_t)g.Cout * g.wtaps * g.Cin * 2 >= (1ll << 31))
        return 0;
    PatchGeo pg;
    // 16 x 16 tiles when the layer still fills the chip with them (two 8-wave workgroups per CU)
    if (!no_256 && c.bn == 128 && (int64_t)(g.M / 256) * (g.Cout / 128) * g.nphase >= 512 && patch_geo(g, 16, &pg)) return 256;
    // 64-wide layers on 16 x 16 tiles with ONE wave column (gg_fwd_patch1_k, four waves of 64 x 64, three workgroups per
    // CU): pays where the reduction is long -- decoders[6] forward (two 128-channel sources, ReLU on load) 170 -> 158 us;
    // the input gradients of encoders[1] / D block 1 (128 channels deep) 91 -> 94 and 188 -> 189: left on the 8 x 16 tile.
    // With two weight-tile buffers (two workgroups per CU) every one of them is 5-25 % slower; two waves of 64 x 64 on the
    // 8 x 16 tile (128-thread workgroups) run those input gradients 13-22 % slower.  tunable fwd_w1 (default 1)
    if (!no_256 && c.bn == 64 && g.Cin >= 256 && pai_tunable("fwd_w1", 1) &&
        (int64_t)(g.M / 256) * (g.Cout / 64) * g.nphase >= pai_tunable("fwd_w1_min", 768) && patch_geo(g, 16, &pg))
        return 256;
    // (measured and dropped: a 16 x 16 tile for the 64-wide layers -- decoders[6], input gradients of encoders[1] /
    // D block 1, whose LDS fill rather than the matrix pipe is the bound, scripts/abl.sh -- ran 10-25 % SLOWER than
    // the 8 x 16 tile at 4-5 workgroups per CU: 185 vs 166 us on decoders[6] forward)
    return patch_geo(g, 8, &pg) ? 128 : 0;
}

// Compile-time timing ablations of gg_fwd_patch_k (results are WRONG; scripts/abl.sh builds the variants):
// 1 no weight-tile fill, 2 no patch fill, 4 no MFMA, 8 no fragment reads, 16 no epilogue
#ifndef PATCH_ABL
#define PATCH_ABL 0
#endif
// timing ablations of the fused-backward store (results WRONG): 1 no z loads, 2 no chunk arithmetic, 4 no stores
#ifndef EPI_ABL
#define EPI_ABL 0
#endif
// (Round 6, built and dropped -- it does not fit the register line: waves 4-7 of the 256-row tile half a step behind waves 0-3 -- the
// second K half of step i - 1 multiplied from fragments fetched BEFORE the barrier of step i (guide, "two waves per SIMD",
// item 9).  The pending fragment set is 32 registers beside 64 accumulators and the 32 of the step in flight: 142+ in a
// 128-register kernel; hipcc spills the fill offsets and reloads them from scratch in front of every weight-tile fill.)
#ifndef PATCH_COLSWZ_ALL
#define PATCH_COLSWZ_ALL 0   // 1: the immediate-offset operand addressing also for the kernels at the 128-register line
#endif
#ifndef PATCH_SETPRIO
#define PATCH_SETPRIO 1   // raise the wave's issue priority over its MFMA cluster (guide T5).  Round 6, same box, interleaved: alone on the chip +1 % time (fused-backward input gradients 1893 -> 1911 us), in the step 5.726 / 5.719 / 5.702 -> 5.701 / 5.702 / 5.675 ms (beside the weight-gradient waves of the side stream)
#endif
// (A v_mfma_f32_32x32x16_bf16 form of this body -- 2 x 2 accumulator tiles of 32 x 32, other LDS swizzle and epilogue
// mapping -- was built and measured in round 2: bit-exact and within +-3 % of this one on every layer, because neither
// issue slots nor the matrix pipe bound the kernel; removed in round 3, numbers in DESIGN.md section 9.)
// Compile-time ablations on decoders[4] (137 GFLOP, forward us): everything 128; no epilogue 120; no patch fill 122; no
// weight-tile fill 106; no LDS-DMA fill at all 91; neither fills nor fragment reads 77 (1.79 PFLOP/s; a bare MFMA loop
// sustains 2.07 on this board with the clock throttled to ~2.07 GHz at the 1400 W power cap): the matrix pipe waits for
// the LDS -- per two co-resident workgroup-steps (1024 matrix cycles per SIMD) the LDS serves 2 x 8 waves x 16
// ds_read_b128 (1024 cycles) plus 2 x 26 KB of LDS-DMA writes.
// WN: waves side by side along the channels.  2: (BM / 64) x 2 waves of 64 pixels x BN / 2 channels.  1 (BN = 64 only):
// BM / 64 waves of 64 pixels x 64 channels -- the 64-channel layers with the 64 x 64 wave tile of the 128-channel
// kernels (8 fragment reads per 16 MFMAs instead of 6 per 8) and one weight tile per 256 pixels.
// WPX: output pixels per wave.  64: 64 x 64 wave tiles (4 + 4 fragment reads per 16 MFMAs).  128: 128 x 64 wave tiles
// (8 + 4 per 32: a quarter fewer LDS fragment bytes per FLOP, the lever that took the weight gradient from 0.75 to 1.0
// PFLOP/s, gg_wg3.hip), 128 accumulator registers, half the waves -- 256 x 128 tiles are then four waves, two
// workgroups per CU at up to 256 registers.
// COLSWZ: the round-3 operand addressing (patch chunks swizzled by the patch COLUMN, taps unrolled: every fragment address
// a register + immediate).  false: the round-1/2 addressing (swizzle by the pixel index, four pixel addresses rebuilt per
// tap, ~30 vector instructions per 32 MFMAs) -- kept for the two configurations of eight / four 64 x 64 waves that sit AT
// their occupancy's register line (128): unrolled, hipcc spills 12-15 registers there and reloads the fill constants from
// scratch in front of every patch fill (same-box step: 6.68-6.74 ms against 6.55-6.59 with this form).
template <int BM, int BN, bool DBB, int WN = 2, int WPX = 64, bool COLSWZ = true>   // DBB: two weight-tile buffers
__device__ __forceinline__ void gg_fwd_patch_body(const GG& g, const FwdArgs& a, const PatchGeo& pg, int mtiles, int ntiles) {
    static_assert(WN == 2 || BN == 64, "one wave column: 64 output channels");
    constexpr int abl = PATCH_ABL;
    typedef PatchDims<BM, WN, WPX> PD;
    constexpr int NTHR = PD::NTHR, MT = WPX / 16, NT = BN / (16 * WN);
    constexpr int BNW = BN / WN;             // channels per wave column
    constexpr int RPP = PD::RPP, PJ = PD::PJ, PATCH_PIX = PD::PIX, PATCH_BYTES = PD::BYTES;
    constexpr int BJ = BN / RPP;             // weight tile fill instructions per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Bs = smem + PATCH_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = WN == 2 ? wid >> 1 : wid, wn = WN == 2 ? wid & 1 : 0;
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int bn = bid % ntiles;
    bid /= ntiles;
    const int ph = bid % g.nphase;
    const int bm = bid / g.nphase;
    const int n0 = bn * BN;
    const int tpi = pg.TY * pg.TX;
    const int img = bm / tpi, trem = bm - img * tpi;
    const int gy0 = (trem / pg.TX) * PD::TH, gx0 = (trem % pg.TX) * 16;

    // Both tiles are filled by LDS-DMA through buffer descriptors (buffer_load_dwordx4 ... lds): the per-lane part of
    // an address is a 32-bit byte offset, the per-step part (tap and channel chunk of the weight tile) rides in the
    // scalar offset, and a lane whose offset lies beyond the buffer writes ZEROS to LDS (scripts/micro/oob_probe.hip):
    // padding pixels need no zero line and no 64-bit select.  host (patch_rows): every tensor is smaller than 2 GB.
    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(a.w), 0, (unsigned)(g.Cout * g.wtaps * g.Cin) * 2u, 0x00020000);
    const unsigned xpix = (unsigned)(g.N * g.H * g.W);
    const __amdgpu_buffer_rsrc_t x1rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(a.x1), 0, xpix * (unsigned)g.C1 * 2u, 0x00020000);
    const __amdgpu_buffer_rsrc_t x2rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(a.x2 ? a.x2 : a.x1), 0, a.x2 ? xpix * (unsigned)g.C2 * 2u : 0u, 0x00020000);
    // window offsets of this phase, 4 bits each (biased by 8): indexing the by-value PatchGeo byte arrays with the
    // run-time window made hipcc read them with vector loads -- and wait for vmcnt(0) -- in front of every patch fill
    unsigned wby16 = 0, wbx16 = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        wby16 |= (unsigned)((pg.by[ph][q] + 8) & 15) << (4 * q);
        wbx16 |= (unsigned)((pg.bx[ph][q] + 8) & 15) << (4 * q);
    }
    wby16 = __builtin_amdgcn_readfirstlane(wby16);
    wbx16 = __builtin_amdgcn_readfirstlane(wbx16);
    // the per-tap tables of this phase in three scalar registers (round 6; packed on the host, patch_geo_pack): weight tap
    // slots 4 bits per (window, tap), patch offsets (ty, tx) 2 bits per (window, tap); plus the ReLU-on-load flags.  Indexed through the by-value structs they were an
    // s_load_dword + s_waitcnt lgkmcnt(0) behind the barrier of EVERY step, in front of the weight-tile fill.
    const unsigned wtlo = __builtin_amdgcn_readfirstlane(pg.wt_lo[ph]);
    const unsigned wthi = __builtin_amdgcn_readfirstlane(pg.wt_hi[ph]);
    const unsigned toff2 = __builtin_amdgcn_readfirstlane(pg.toff2[ph]);
    const unsigned relu_bits = __builtin_amdgcn_readfirstlane((g.relu1 ? 1u : 0u) | (g.relu2 ? 2u : 0u));

    // ---- patch fill map: thread -> (pixel p = 32 j + tid / 8, 16-B slot tid % 8) ----------------
    const int sc = lane & 7, sr = wid * 8 + (lane >> 3);
    // slot c of patch pixel (py, px) holds chunk c ^ (px & 6): the swizzle looks at the patch COLUMN only, so the pixel
    // rows a wave reads (its four 16-pixel rows, shifted down by the tap's ty) are ONE base address + immediates
    // (scripts/lds_swizzle_check.py fwd_patch: conflict-free for both tx shifts)
    // per fill instruction ONE register: bits 0-23 source pixel index of the patch pixel for window offset (0, 0) (every
    // tensor is below 2 GB and has >= 64 channels: < 2^24 pixels), bits 24-27 "inside the image" for window q of this
    // phase, bits 28-30 this thread's chunk.  (The kernel sits at the 128-register line of four waves per SIMD.)
    unsigned pfill[PJ];
#pragma unroll
    for (int j = 0; j < PJ; ++j) {
        const int p = j * RPP + sr;
        const int py = p / PATCH_W, px = p - py * PATCH_W;
        const int y = (gy0 + py) * g.S, x = (gx0 + px) * g.S;
        const int pixb = (img * g.H + y) * g.W + x;
        unsigned m = 0;
        auto inside = [&](int q) {
            const int yy = y + (int)((wby16 >> (4 * q)) & 15u) - 8, xx = x + (int)((wbx16 >> (4 * q)) & 15u) - 8;
            if (p < PATCH_PIX && (unsigned)yy < (unsigned)g.H && (unsigned)xx < (unsigned)g.W) m |= 1u << q;
        };
        inside(0);
        if (pg.groups > 1) {   // (wave-uniform: the one-window phases of the transposed forms skip three quarters of this)
#pragma unroll
            for (int q = 1; q < 4; ++q) inside(q);
        }
        pfill[j] = ((unsigned)pixb & 0xffffffu) | (m << 24) | ((unsigned)(sc ^ ((COLSWZ ? px : sr) & 6)) << 28);
    }
    const int gchB = (sc ^ ((sr >> 1) & 7)) * 8;
    // LDS row rho = 16 nt + i of a wave's half of the weight tile holds output channel
    // (4 NT) (i >> 2) + 4 nt + (i & 3) of that half: with the weights as the MFMA's A operand a lane then ends
    // up with 4 NT CONSECUTIVE channels of one pixel, and the epilogue stages 16-B pieces instead of 2-B ones
    unsigned wrow[BJ];                       // byte offset of this lane's chunk of weight row j (tap 0, channel 0)
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        const int lr = sr + RPP * j, half = lr / BNW, rho = lr % BNW;
        const int ch = half * BNW + (4 * NT) * ((rho & 15) >> 2) + 4 * (rho >> 4) + (rho & 3);
        wrow[j] = (unsigned)((n0 + ch) * g.wtaps * g.Cin + gchB) * 2u;
    }

    // ---- fragment read addresses -----------------------------------------------------------------
    // Every fragment address of the K loop is one of these six registers + a compile-time immediate: the patch offset
    // (ty, tx) of tap slot k is (k >> 1, k & 1) by construction (patch_geo), so ty and the wave's pixel row mt are
    // immediates ((mt + ty) x 17 pixels), and so are the weight buffer (k & 1) and channel tile nt.  (Until round 3 the
    // loop rebuilt four pixel addresses per tap from a run-time offset: ~30 vector instructions per 32 MFMAs.)
    const int fr = lane & 15, fq = lane >> 4;
    unsigned a_addr[2][2];                   // [tx][kk]
#pragma unroll
    for (int tx = 0; tx < 2; ++tx)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const unsigned px = (unsigned)(fr + tx);
            a_addr[tx][kk] = ((unsigned)(wm * MT * PATCH_W) + px) * 128u + ((((unsigned)(kk * 4 + fq)) ^ (px & 6u)) << 4);
        }
    const int fswz = fr >> 1;
    unsigned b_addr[2];                      // [kk]
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
        b_addr[kk] = (unsigned)(PATCH_BYTES + (wn * BNW + fr) * 128) + (unsigned)(((kk * 4 + fq) ^ fswz) << 4);
    int pbase[MT];                           // !COLSWZ: patch pixel of the wave's row mt (tap offset added per step)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) pbase[mt] = (wm * MT + mt) * PATCH_W + fr;
    const unsigned b_base = (unsigned)(PATCH_BYTES + (wn * BNW + fr) * 128);
    f4_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f4_t){0.f, 0.f, 0.f, 0.f};

    // One step = one tap of one window: 64 channels of the patch against one weight tile.  The weight tile
    // of step i+1 is in flight (second buffer) while step i is multiplied; the patch is replaced every 4
    // steps, behind a barrier of its own.
    const int cchunks = g.Cin / MBK;
    const int ngroups = cchunks * pg.groups;
    const int gsh = pg.groups == 4 ? 2 : 0;
#define FP_BLDS16(rs, voff, soff, lptr) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lptr), 16, (int)(voff), (int)(soff), 0, 0)
    auto fire_patch = [&](int gi) {
        const int c0 = (gi >> gsh) * MBK, q = gi & (pg.groups - 1);
        const bool second = c0 >= g.C1;
        const int C = second ? g.C2 : g.C1;
        const int cofs = second ? c0 - g.C1 : c0;
        const int dpix = ((int)((wby16 >> (4 * q)) & 15u) - 8) * g.W + (int)((wbx16 >> (4 * q)) & 15u) - 8;
        if (abl & 2) return;
#pragma unroll
        for (int j = 0; j < PJ; ++j) {
            // (opaque copy: hipcc otherwise hoists the three fields of every pfill[j] out of the K loop as fifteen
            //  loop invariants and spills them -- five scratch round trips in front of every patch fill)
            unsigned pf = pfill[j];
            asm volatile("" : "+v"(pf));
            const unsigned vo = ((pf >> (24 + q)) & 1u)
                                    ? (unsigned)(((int)(pf & 0xffffffu) + dpix) * C + cofs + (int)((pf >> 28) & 7u) * 8) * 2u : OOB;
            if (second) FP_BLDS16(x2rs, vo, 0, smem + (j * RPP + wid * 8) * 128);
            else FP_BLDS16(x1rs, vo, 0, smem + (j * RPP + wid * 8) * 128);
        }
    };
    auto fire_b = [&](int gi, int k, int buf) {
        const int c0 = (gi >> gsh) * MBK, q = gi & (pg.groups - 1);
        const unsigned slot = (((q & 2) ? wthi : wtlo) >> (16 * (q & 1) + 4 * k)) & 15u;
        const unsigned woff = (unsigned)((int)slot * g.Cin + c0) * 2u;
        if (abl & 1) return;
#pragma unroll
        for (int j = 0; j < BJ; ++j) FP_BLDS16(wrs, wrow[j], woff, Bs + buf * (BN * 128) + (j * RPP + wid * 8) * 128);
    };
    fire_patch(0);
    fire_b(0, 0, 0);
    if constexpr (COLSWZ) {
        // one tap (slot K of the window, patch offset (K >> 1, K & 1)); the weight buffer of step K is K & 1 (four steps per
        // patch: the parity restarts with every patch)
        auto step = [&](auto k_tag, int gi, int relu, bool more) {
            constexpr int K = decltype(k_tag)::value;
            constexpr int TY = K >> 1, TX = K & 1;
            constexpr int BUF = DBB ? (K & 1) : 0;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();   // this step's tiles have landed; everyone is done with the other weight buffer
            if (DBB) {
                if (K < 3) fire_b(gi, K + 1, BUF ^ 1);
                else if (more) fire_b(gi + 1, 0, BUF ^ 1);
            }
    #pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf8_t af[MT], bfr[NT];
                if (!(abl & 8)) {
                    const unsigned aa = a_addr[TX][kk], ba = b_addr[kk];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        af[mt] = *(const bf8_t*)(smem + (aa & 0xffffu) + (mt + TY) * (PATCH_W * 128));
    #pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        bfr[nt] = *(const bf8_t*)(smem + (ba & 0x3ffffu) + BUF * (BN * 128) + nt * 16 * 128);
                } else {
    #pragma unroll
                    for (int mt = 0; mt < MT; ++mt) af[mt] = __builtin_bit_cast(bf8_t, make_uint4(a_addr[TX][kk], b_addr[kk], mt, kk));
    #pragma unroll
                    for (int nt = 0; nt < NT; ++nt) bfr[nt] = __builtin_bit_cast(bf8_t, make_uint4(b_addr[kk], a_addr[TX][kk], nt, kk));
                }
                if (relu) {
    #pragma unroll
                    for (int mt = 0; mt < MT; ++mt) af[mt] = relu_frag(af[mt]);
                }
                if (PATCH_SETPRIO) __builtin_amdgcn_s_setprio(1);
                if (!(abl & 4)) {
    #pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
    #pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            // D[i = channel slot][j = pixel]: acc[mt][nt][r] = channel slot 4 fq + r of pixel fr
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[nt], af[mt], acc[mt][nt], 0, 0, 0);
                } else {
                    acc[0][0][0] += (float)af[0][0] + (float)bfr[0][0];
                }
                if (PATCH_SETPRIO) __builtin_amdgcn_s_setprio(0);
            }
            if (DBB) {
                if (K == 3 && more) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();   // every wave is done reading the patch
                    fire_patch(gi + 1);
                }
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();       // every wave is done reading before the next fill overwrites
                if (K < 3) fire_b(gi, K + 1, 0);
                else if (more) { fire_patch(gi + 1); fire_b(gi + 1, 0, 0); }
            }
        };
        for (int gi = 0; gi < ngroups; ++gi) {
            const int c0g = (gi >> gsh) * MBK;
            const int relu = (int)((relu_bits >> (c0g >= g.C1 ? 1 : 0)) & 1u);
            const bool more = gi + 1 < ngroups;
            step(std::integral_constant<int, 0>{}, gi, relu, more);
            step(std::integral_constant<int, 1>{}, gi, relu, more);
            step(std::integral_constant<int, 2>{}, gi, relu, more);
            step(std::integral_constant<int, 3>{}, gi, relu, more);
        }
    } else {
        int buf = 0;
        for (int gi = 0; gi < ngroups; ++gi) {
            const int c0g = (gi >> gsh) * MBK;
            const int relu = (int)((relu_bits >> (c0g >= g.C1 ? 1 : 0)) & 1u);
            const unsigned toff8 = toff2 >> (8 * (gi & (pg.groups - 1)));
            const bool more = gi + 1 < ngroups;
    #pragma unroll 1
            for (int k = 0; k < 4; ++k) {
                const int t2 = (int)((toff8 >> (2 * k)) & 3u), toff = (t2 >> 1) * PATCH_W + (t2 & 1);
                unsigned abase[MT];
    #pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const unsigned pp = (unsigned)(pbase[mt] + toff);
                    abase[mt] = (pp << 7) ^ ((pp & 6u) << 4);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();   // this step's tiles have landed; everyone is done with the other weight buffer
                if (DBB) {
                    if (k < 3) fire_b(gi, k + 1, buf ^ 1);
                    else if (more) fire_b(gi + 1, 0, buf ^ 1);
                }
                const unsigned bb = b_base + (DBB ? buf * (BN * 128) : 0);
    #pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const unsigned ca = (unsigned)((kk * 4 + fq) << 4);
                    const unsigned cb = (unsigned)(((kk * 4 + fq) ^ fswz) << 4);
                    bf8_t af[MT], bfr[NT];
                    if (!(abl & 8)) {
    #pragma unroll
                        for (int mt = 0; mt < MT; ++mt) af[mt] = *(const bf8_t*)(smem + (abase[mt] ^ ca));
    #pragma unroll
                        for (int nt = 0; nt < NT; ++nt) bfr[nt] = *(const bf8_t*)(smem + bb + nt * 16 * 128 + cb);
                    } else {
    #pragma unroll
                        for (int mt = 0; mt < MT; ++mt) af[mt] = __builtin_bit_cast(bf8_t, make_uint4(ca, cb, mt, kk));
    #pragma unroll
                        for (int nt = 0; nt < NT; ++nt) bfr[nt] = __builtin_bit_cast(bf8_t, make_uint4(cb, ca, nt, kk));
                    }
                    if (relu) {
    #pragma unroll
                        for (int mt = 0; mt < MT; ++mt) af[mt] = relu_frag(af[mt]);
                    }
                    if (PATCH_SETPRIO) __builtin_amdgcn_s_setprio(1);
                    if (!(abl & 4)) {
    #pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
    #pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                // D[i = channel slot][j = pixel]: acc[mt][nt][r] = channel slot 4 fq + r of pixel fr
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[nt], af[mt], acc[mt][nt], 0, 0, 0);
                    } else {
                        acc[0][0][0] += (float)af[0][0] + (float)bfr[0][0];
                    }
                    if (PATCH_SETPRIO) __builtin_amdgcn_s_setprio(0);
                }
                if (DBB) {
                    buf ^= 1;
                    if (k == 3 && more) {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();   // every wave is done reading the patch
                        fire_patch(gi + 1);
                    }
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();       // every wave is done reading before the next fill overwrites
                    if (k < 3) fire_b(gi, k + 1, 0);
                    else if (more) { fire_patch(gi + 1); fire_b(gi + 1, 0, 0); }
                }
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // the epilogue reuses the tile memory
    if (abl & 16) {   // every accumulator stays live, nothing of the epilogue runs
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
            }
        if (t == 123.456f) *(float*)a.y1 = t;
        return;
    }

    // ---- epilogue: bias, BN partial statistics, activation, LDS-staged row stores --------------
    // No run-time switch per element (profiles/r06_isa_census.txt): bias / statistics / activation are wave-uniform
    // branches around 16 elements, the activation is act_fwd (gg_tile.h), and the store loop below is compiled once per
    // (producer backward, skip gradient, BatchNorm sums) combination.
    constexpr int CROW = BN * 2 + 16;
    constexpr int WM = BM / WPX;
    unsigned char* Cs = smem;
    float* sstat = (float*)(smem + BM * CROW);  // [WM][2][BN]
    const int eact = a.yact ? a.eact : PAI_ACT_NONE;
    const float eslope = act_slope(eact);
    // lane (fq, fr) holds, for each of its 4 pixel rows mt*16 + fr, the 4 NT consecutive channels
    // wn*(BN/2) + 4 NT fq + (4 nt + r): bias, statistics, activation, then one or two 16-B LDS stores per row
    constexpr int CL = 4 * NT;               // channels per lane
    const int col0 = wn * BNW + CL * fq;
    bf16_t* dst;
    int dstride, dcol;
    if (a.yact) { dst = (bf16_t*)a.yact; dstride = g.Cout; dcol = n0; }
    else if (n0 < g.D1) { dst = (bf16_t*)a.y1; dstride = g.D1; dcol = n0; }
    else { dst = (bf16_t*)a.y2; dstride = g.D2; dcol = n0 - g.D1; }
    const bool bwd = a.bz && !a.yact && n0 < g.D1;   // uniform per workgroup
    const bf16_t* bzp = (const bf16_t*)a.bz;
    const bf16_t* bap = (const bf16_t*)a.badd;
    const bool bsum = bwd && a.bpart;
    constexpr int CPR = BN / 8;        // 16-B chunks per row
    constexpr int ORP = NTHR / CPR;    // rows per pass
    static_assert(ORP % 16 == 0, "a pass covers whole 16-pixel tile rows");
    const int oc = tid % CPR, orow0 = tid / CPR;
    constexpr int NP = BM / ORP, NB = NP < 4 ? NP : 4, NBATCH = NP / NB;
    // pass p of this thread: tile row orow0 + p ORP, i.e. ORP / 16 image rows further down.  Addresses = a 64-bit base that is
    // uniform over the workgroup (scalar registers) + a 32-bit byte offset per thread + a uniform step per pass.
    const size_t tile0 = ((size_t)(img * g.OH + gy0 * g.OS + g.poy[ph]) * g.OW + gx0 * g.OS + g.pox[ph]) * dstride + dcol;
    const unsigned toff = (unsigned)((((orow0 >> 4) * g.OS * g.OW + (orow0 & 15) * g.OS) * dstride + oc * 8) * 2);
    const unsigned pstep = (unsigned)((ORP / 16) * g.OS * g.OW * dstride * 2);
    const char* zt = (const char*)(bzp + tile0);
    const char* at = (const char*)(bap + tile0);
    char* dt = (char*)(dst + tile0);
    const float sl1 = act_slope(a.bact1), sl2 = act_slope(a.bact2);
    // The first NB passes of the producer's chunks (fused backward) are requested HERE, ahead of the staging pass and its
    // barrier: their HBM latency runs beside the staging instead of behind it.
    // (the skip-gradient chunks follow behind the staging pass, when the accumulators' registers are free: all eight requests
    //  up front did not fit beside 64 accumulators at the 128-register line)
    uint4 zq[NB], aq[NB];
    if (bwd && !(EPI_ABL & 1)) {
#pragma unroll
        for (int p = 0; p < NB; ++p) zq[p] = *(const uint4*)(zt + (toff + (unsigned)p * pstep));
    }
    const bool has_bias = a.bias != nullptr, has_stats = a.stats != nullptr;
    // two forms of the staging pass: bare (input gradients: neither bias nor statistics, no register arrays for them) and full
    auto stage = [&](auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        float bias_v[FULL ? CL : 1], csum[FULL ? CL : 1], csq[FULL ? CL : 1];
        if (FULL) {
#pragma unroll
            for (int c = 0; c < CL; ++c) { bias_v[c] = has_bias ? a.bias[n0 + col0 + c] : 0.f; csum[c] = csq[c] = 0.f; }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int row = wm * WPX + mt * 16 + fr;
#pragma unroll
            for (int h = 0; h < CL / 8; ++h) {
                float v[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    v[c] = acc[mt][(8 * h + c) >> 2][c & 3];
                    if (FULL) {
                        v[c] += bias_v[8 * h + c];
                        csum[8 * h + c] += v[c];
                        csq[8 * h + c] = fmaf(v[c], v[c], csq[8 * h + c]);
                    }
                }
                if (eact != PAI_ACT_NONE) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) v[c] = act_fwd(v[c], eslope);
                }
                *(uint4*)(Cs + row * CROW + (col0 + 8 * h) * 2) =
                    make_uint4(pk2bf(v[0], v[1]), pk2bf(v[2], v[3]), pk2bf(v[4], v[5]), pk2bf(v[6], v[7]));
            }
        }
        if (FULL && has_stats) {
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
    };
    if (has_bias || has_stats) stage(std::true_type{});
    else stage(std::false_type{});
    if (bwd && bap) {
#pragma unroll
        for (int p = 0; p < NB; ++p) aq[p] = *(const uint4*)(at + (toff + (unsigned)p * pstep));
    }
    // (behind the staging pass: at the 128-register line the 32 registers of parameters and sums beside the accumulators
    //  and the chunks in flight made hipcc wait for every chunk and park it in scratch)
    BwdParams BP;
    float bs1[8], bs2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) bs1[k] = bs2[k] = 0.f;
    if (bwd) bwd_load_params(a, dcol + oc * 8, BP);
    __syncthreads();
    if (has_stats && tid < BN) {
        float* dst = a.stats + ((size_t)(ph * mtiles + bm) * 2) * g.Cout + n0 + tid;
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int i = 0; i < WM; ++i) { s += sstat[(i * 2 + 0) * BN + tid]; q += sstat[(i * 2 + 1) * BN + tid]; }
        dst[0] = s;
        dst[g.Cout] = q;
    }
    // The chunks of the producer's tensors are requested four passes ahead of the stores they feed: the stores may alias them
    // as far as the compiler knows, so the order is written out -- a pass's registers are re-requested for pass + 4 as soon
    // as its chunk is computed.
    auto store_tile = [&](auto bwd_tag, auto add_tag, auto sum_tag) {
        constexpr bool BWD = decltype(bwd_tag)::value, ADD = decltype(add_tag)::value, SUM = decltype(sum_tag)::value;
        auto request = [&](int pass, int slot) {
            const unsigned off = toff + (unsigned)pass * pstep;
            if (BWD && !(EPI_ABL & 1)) zq[slot] = *(const uint4*)(zt + off);
            if (BWD && ADD) aq[slot] = *(const uint4*)(at + off);
        };
#pragma unroll
        for (int b = 0; b < NBATCH; ++b) {
#pragma unroll
            for (int p = 0; p < NB; ++p) {
                const int pass = b * NB + p;
                uint4 o = *(const uint4*)(Cs + (orow0 + pass * ORP) * CROW + oc * 16);
                if (EPI_ABL & 1) zq[p] = o;
                if (BWD && !(EPI_ABL & 2)) o = bwd_chunk_t<ADD, SUM>(o, zq[p], aq[p], sl1, sl2, BP, bs1, bs2);
                if (BWD && (EPI_ABL & 2)) { o.x ^= zq[p].x; if (ADD) o.y ^= aq[p].y; }
                if (b + 1 < NBATCH) request(pass + NB, p);   // rolling window: NB passes of producer chunks in flight
                if (!(EPI_ABL & 4) || o.x == 0x12345u) *(uint4*)(dt + (toff + (unsigned)pass * pstep)) = o;
                // one pass at a time: left alone, hipcc unpacks the chunks of all four passes in flight up front (~100 live
                // registers of floats) and spills the parameters and sums around them
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    typedef std::true_type T_;
    typedef std::false_type F_;
    if (!bwd) store_tile(F_{}, F_{}, F_{});
    else if (bap) { if (bsum) store_tile(T_{}, T_{}, T_{}); else store_tile(T_{}, T_{}, F_{}); }
    else { if (bsum) store_tile(T_{}, F_{}, T_{}); else store_tile(T_{}, F_{}, F_{}); }
    if (bsum)
        bwd_write_partials<BN, CPR, NTHR / 64>(sstat, bs1, bs2, tid,
                                               a.bpart + ((size_t)(ph * mtiles + bm) * 2) * g.D1 + n0, g.D1,
                                               a.bmean + n0, a.brstd + n0);
}

template <int BM, int BN, bool DBB>
__global__ __launch_bounds__(BM * 2, (BM == 128 && DBB) ? (BN == 64 ? 4 : 3) : (BN == 64 ? 5 : 4)) void gg_fwd_patch_k(GG g, FwdArgs a, PatchGeo pg, int mtiles, int ntiles) {
    // (BM, BN, DBB) = (256, 128, *) and (128, 128, false) are the configurations at the register line, see COLSWZ
    gg_fwd_patch_body<BM, BN, DBB, 2, 64, PATCH_COLSWZ_ALL || !(BN == 128 && (BM == 256 || !DBB))>(g, a, pg, mtiles, ntiles);
}
// 64 output channels, one wave column: BM / 64 waves of 64 x 64 (see gg_fwd_patch_body, WN = 1)
template <int BM, int BN, bool DBB>
__global__ __launch_bounds__(BM, 3) void gg_fwd_patch1_k(GG g, FwdArgs a, PatchGeo pg, int mtiles, int ntiles) {
    gg_fwd_patch_body<BM, BN, DBB, 1>(g, a, pg, mtiles, ntiles);
}
template <int BM, int BN, bool DB, int WR = 64>
static size_t fwd_lds_bytes() {
    const size_t main_loop = (size_t)(DB ? 2 : 1) * (BM * 128 + BN * 128);
    const size_t epilogue = BM * (BN * 2 + 16) + (BM / WR * 2) * 2 * BN * sizeof(float);   // staged tile + [waves][2][BN]
    return main_loop > epilogue ? main_loop : epilogue;
}

int launch_fwd_mfma(const GG& g, const FwdArgs& a, hipStream_t s) {
    FwdCfg c = fwd_cfg(g);
    c.ksplit = fwd_effective_ksplit(g);
    const int mtiles = cdiv(g.M, abs(c.bm));
    const int ntiles = g.Cout / c.bn;
    const dim3 grid(mtiles * ntiles * g.nphase * c.ksplit);
#define FWD_LAUNCH(BM, BN, SK, DB)                                                                  \
    PAI_LAUNCH((gg_fwd_mfma_k<BM, BN, SK, DB>), grid, dim3(BM * 2), (fwd_lds_bytes<BM, BN, DB>()), s, g, \
                       a, mtiles, ntiles, c.ksplit, pai_ctx()->workspace)
    if (c.bm == 256) {
        static PerDeviceOnce attr_set;   // > 64 KB of dynamic LDS needs an explicit opt-in
        if (attr_set.first()) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gg_fwd_mfma_k<256, 128, false, true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)fwd_lds_bytes<256, 128, true>());
            PAI_CHECK(e == hipSuccess, "hipFuncSetAttribute(max dynamic LDS): %s", hipGetErrorString(e));
        }
        FWD_LAUNCH(256, 128, false, true);
    } else if (c.bm == -128) {
        FWD_LAUNCH(128, 128, false, true);
    } else if (c.bm == 128 && c.bn == 128 && c.ksplit == 1 && getenv("PAI_FWD_MODE") && atoi(getenv("PAI_FWD_MODE")) == 3) {
        PAI_LAUNCH((gg_fwd_mfma_k<128, 128, false, false, 32>), grid, dim3(512),
                           (fwd_lds_bytes<128, 128, false, 32>()), s, g, a, mtiles, ntiles, c.ksplit, pai_ctx()->workspace);
    } else if (c.ksplit > 1) {
        if (c.bn == 128) {
            // two LDS stages (counted vmcnt + raw barrier): these launches have few workgroups and a long K loop per
            // workgroup, i.e. nobody else hides their L2 -> LDS latency (encoders[4] forward 50 -> 45 us, decoders[3]
            // input gradient 83 -> 76 us, scripts/micro/convbench)
            if (pai_tunable("fwd_splitk_db", 1)) FWD_LAUNCH(128, 128, true, true); else FWD_LAUNCH(128, 128, true, false);
        } else FWD_LAUNCH(128, 64, true, false);
        PAI_LAUNCH_CHECK();
        const int ftiles = cdiv(g.M, FIN_ROWS);
        PAI_LAUNCH(splitk_finish_k, dim3(ftiles, g.nphase, cdiv(g.Cout, FIN_COLS)), dim3(256), 0, s, g, a,
                           pai_ctx()->workspace, ftiles, c.ksplit);
    } else {
        PatchGeo pg;
        const int prow = patch_rows(g, c);
        // second weight-tile buffer: pays on the 128-wide tiles (bit 0: 256-row, bit 1: 128-row), not on the
        // 64-wide ones (bit 2), whose 8 KB weight tile is cheap to wait for and which lose a workgroup per CU to it
        static const int dbb = getenv("PAI_PATCH_DBB") ? atoi(getenv("PAI_PATCH_DBB")) : 3;
        if (prow == 256 && c.bn == 64 && patch_geo(g, 16, &pg)) {
            typedef PatchDims<256, 1> PD;
            const bool db = (dbb & 4) != 0;
            const size_t lds = PD::BYTES + (size_t)64 * 128 * (db ? 2 : 1);
            const size_t epi = 256 * ((size_t)64 * 2 + 16) + 4 * 2 * 64 * sizeof(float);
            const size_t need = lds > epi ? lds : epi;
            const int mt256 = g.M / 256;
            const dim3 grid256(mt256 * ntiles * g.nphase);
            if (db) PAI_LAUNCH((gg_fwd_patch1_k<256, 64, true>), grid256, dim3(256), need, s, g, a, pg, mt256, ntiles);
            else PAI_LAUNCH((gg_fwd_patch1_k<256, 64, false>), grid256, dim3(256), need, s, g, a, pg, mt256, ntiles);
        } else if (prow == 256 && patch_geo(g, 16, &pg)) {
            typedef PatchDims<256> PD;
            const bool db = (dbb & 1) != 0;
            const size_t lds = PD::BYTES + (size_t)128 * 128 * (db ? 2 : 1);
            const size_t epi = 256 * ((size_t)128 * 2 + 16) + 8 * 2 * 128 * sizeof(float);
            const size_t need = lds > epi ? lds : epi;
            static PerDeviceOnce attr;
            if (attr.first()) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gg_fwd_patch_k<256, 128, true>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
                if (e == hipSuccess)
                    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gg_fwd_patch_k<256, 128, false>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
                PAI_CHECK(e == hipSuccess, "hipFuncSetAttribute(max dynamic LDS): %s", hipGetErrorString(e));
            }
            const int mt256 = g.M / 256;
            const dim3 grid256(mt256 * ntiles * g.nphase);
            if (db) PAI_LAUNCH((gg_fwd_patch_k<256, 128, true>), grid256, dim3(512), need, s, g, a, pg, mt256, ntiles);
            else PAI_LAUNCH((gg_fwd_patch_k<256, 128, false>), grid256, dim3(512), need, s, g, a, pg, mt256, ntiles);
        } else if (prow == 128 && patch_geo(g, 8, &pg)) {
            typedef PatchDims<128> PD;
            const bool db = (dbb & (c.bn == 128 ? 2 : 4)) != 0;
            const size_t lds = PD::BYTES + (size_t)c.bn * 128 * (db ? 2 : 1);
            const size_t epi = 128 * ((size_t)c.bn * 2 + 16) + 4 * 2 * c.bn * sizeof(float);
            const size_t need = lds > epi ? lds : epi;
            if (c.bn == 128) {
                if (db) PAI_LAUNCH((gg_fwd_patch_k<128, 128, true>), grid, dim3(256), need, s, g, a, pg, mtiles, ntiles);
                else PAI_LAUNCH((gg_fwd_patch_k<128, 128, false>), grid, dim3(256), need, s, g, a, pg, mtiles, ntiles);
            } else {
                if (db) PAI_LAUNCH((gg_fwd_patch_k<128, 64, true>), grid, dim3(256), need, s, g, a, pg, mtiles, ntiles);
                else PAI_LAUNCH((gg_fwd_patch_k<128, 64, false>), grid, dim3(256), need, s, g, a, pg, mtiles, ntiles);
            }
        } else if (c.bn == 128) FWD_LAUNCH(128, 128, false, false); else FWD_LAUNCH(128, 64, false, false);
    }
#undef FWD_LAUNCH
    PAI_LAUNCH_CHECK();
    return 0;
}

// rocprofv3-visible symbol of the main kernel launch_fwd_mfma picks for this problem (same decisions, no launch)
const char* fwd_mfma_kernel_name(const GG& g) {
    FwdCfg c = fwd_cfg(g);
    c.ksplit = fwd_effective_ksplit(g);
    const int mode = getenv("PAI_FWD_MODE") ? atoi(getenv("PAI_FWD_MODE")) : 0;
    if (c.bm == 256) return "gg_fwd_mfma_k<256, 128, false, true, 64>";
    if (c.bm == -128) return "gg_fwd_mfma_k<128, 128, false, true, 64>";
    if (c.bm == 128 && c.bn == 128 && c.ksplit == 1 && mode == 3) return "gg_fwd_mfma_k<128, 128, false, false, 32>";
    if (c.ksplit > 1)
        return c.bn == 128 ? (pai_tunable("fwd_splitk_db", 1) ? "gg_fwd_mfma_k<128, 128, true, true, 64>" : "gg_fwd_mfma_k<128, 128, true, false, 64>")
                           : "gg_fwd_mfma_k<128, 64, true, false, 64>";
    const int dbb = getenv("PAI_PATCH_DBB") ? atoi(getenv("PAI_PATCH_DBB")) : 3;
    const int prow = patch_rows(g, c);
    if (prow == 256 && c.bn == 64) return (dbb & 4) ? "gg_fwd_patch1_k<256, 64, true>" : "gg_fwd_patch1_k<256, 64, false>";
    if (prow == 256) return (dbb & 1) ? "gg_fwd_patch_k<256, 128, true>" : "gg_fwd_patch_k<256, 128, false>";
    if (prow == 128) {
        if (c.bn == 128) return (dbb & 2) ? "gg_fwd_patch_k<128, 128, true>" : "gg_fwd_patch_k<128, 128, false>";
        return (dbb & 4) ? "gg_fwd_patch_k<128, 64, true>" : "gg_fwd_patch_k<128, 64, false>";
    }
    return c.bn == 128 ? "gg_fwd_mfma_k<128, 128, false, false, 64>" : "gg_fwd_mfma_k<128, 64, false, false, 64>";
}

// ------------------------------------------------------------------------------------
// Skinny pointwise convolution (forward and input gradient of the finest attention gate)
// ------------------------------------------------------------------------------------
// out[M][COUT] = in[M][CIN] x W[COUT][CIN]^T with (CIN, COUT) = (64, 32) or (32, 64) over ~1 M pixels: 4 GFLOP
// against 200 MB, i.e. HBM-bound.  No LDS: a wave keeps the whole filter in registers as the MFMA's A operand
// (rows permuted as in gg_fwd_patch_k, so that a lane ends up with COUT/4 consecutive channels of one pixel) and
// streams 16-pixel groups: one 16-B load per lane and 32 input channels, COUT/16 * CIN/32 MFMAs, one or two 16-B
// stores per lane -- the 16 pixels of a group are contiguous, every load / store instruction covers whole rows.
// Same epilogue options as the tile kernels: bias, BatchNorm partial statistics, the fused producer backward.
// (Round 3, measured and dropped: the next group's loads issued before the current group's MFMAs + a forward-only
//  instantiation without the 80 registers of the fused backward (216 -> 148): the gate's input-gradient launch stayed at
//  164 us and the Attention U-Net step at 8.68-8.74 ms, three interleaved runs.)
bool pw_ok(int dtype, const GG& g, const FwdArgs& a) {
    if (dtype != PAI_BF16 || g.ntaps != 1 || g.nphase != 1 || g.C2 != 0 || g.D2 != 0) return false;
    if (!((g.C1 == 64 && g.Cout == 32) || (g.C1 == 32 && g.Cout == 64))) return false;
    if (a.yf32 || a.skip_d1) return false;
    if ((a.y1 != nullptr) == (a.yact != nullptr)) return false;     // exactly one storage-dtype output
    if (a.yact && a.eact != PAI_ACT_NONE && a.eact != PAI_ACT_LRELU && a.eact != PAI_ACT_RELU) return false;
    return true;
}

int pw_rows(const GG& g) {
    int64_t b = ((int64_t)g.M + 63) / 64;
    return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

template <int CIN, int COUT>
__global__ __launch_bounds__(256) void pw_k(GG g, FwdArgs a, int groups_per_wave) {
    constexpr int KB = CIN / 32, NTT = COUT / 16, CL = COUT / 4, NCH = CL / 8;
    __shared__ float sred[4][2][COUT];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const bf16_t* x = (const bf16_t*)a.x1;
    const bf16_t* w = (const bf16_t*)a.w;
    bf16_t* y = (bf16_t*)(a.yact ? a.yact : a.y1);
    const float eslope = act_slope(a.yact ? a.eact : PAI_ACT_NONE);
    // filter: MFMA row (nt, i = fr) carries output channel CL (i >> 2) + 4 nt + (i & 3)
    bf8_t wf[NTT][KB];
#pragma unroll
    for (int nt = 0; nt < NTT; ++nt)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
            wf[nt][kb] = *(const bf8_t*)(w + (size_t)(CL * (fr >> 2) + 4 * nt + (fr & 3)) * CIN + kb * 32 + fq * 8);
    const int c0 = CL * fq;                  // this lane's first output channel
    float bias_v[CL], s1[CL], s2[CL];
#pragma unroll
    for (int c = 0; c < CL; ++c) { bias_v[c] = a.bias ? a.bias[c0 + c] : 0.f; s1[c] = s2[c] = 0.f; }
    const bool bwd = a.bz != nullptr && !a.yact;
    const bool bsum = bwd && a.bpart;
    const bf16_t* bzp = (const bf16_t*)a.bz;
    const bf16_t* bap = (const bf16_t*)a.badd;
    BwdParams BP[NCH];
    if (bwd) {
#pragma unroll
        for (int h = 0; h < NCH; ++h) bwd_load_params(a, c0 + 8 * h, BP[h]);
    }
    const int64_t ngroups = ((int64_t)g.M + 15) / 16;
    const int64_t g0 = ((int64_t)blockIdx.x * 4 + wid) * groups_per_wave;
    for (int64_t gi = g0; gi < g0 + groups_per_wave && gi < ngroups; ++gi) {
        const int64_t pix = gi * 16 + fr;
        const bool valid = pix < g.M;
        const int64_t pc = valid ? pix : 0;
        bf8_t xb[KB];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            xb[kb] = *(const bf8_t*)(x + pc * CIN + kb * 32 + fq * 8);
            if (g.relu1) xb[kb] = relu_frag(xb[kb]);
        }
        uint4 zq[NCH], aq[NCH];
        if (bwd) {
#pragma unroll
            for (int h = 0; h < NCH; ++h) {
                zq[h] = *(const uint4*)(bzp + pc * COUT + c0 + 8 * h);
                aq[h] = bap ? *(const uint4*)(bap + pc * COUT + c0 + 8 * h) : make_uint4(0, 0, 0, 0);
            }
        }
        f4_t acc[NTT];
#pragma unroll
        for (int nt = 0; nt < NTT; ++nt) {
            acc[nt] = (f4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt][kb], xb[kb], acc[nt], 0, 0, 0);
        }
        // lane: pixel `pix`, channels c0 + 4 nt + r
        unsigned pk[CL / 2];
#pragma unroll
        for (int nt = 0; nt < NTT; ++nt) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = acc[nt][r] + bias_v[4 * nt + r];
                if (a.stats && valid) { s1[4 * nt + r] += v[r]; s2[4 * nt + r] = fmaf(v[r], v[r], s2[4 * nt + r]); }
                v[r] = act_fwd(v[r], eslope);
            }
            pk[2 * nt] = pk2bf(v[0], v[1]);
            pk[2 * nt + 1] = pk2bf(v[2], v[3]);
        }
#pragma unroll
        for (int h = 0; h < NCH; ++h) {
            uint4 o = make_uint4(pk[4 * h], pk[4 * h + 1], pk[4 * h + 2], pk[4 * h + 3]);
            if (bwd && valid)
                o = bwd_chunk(o, zq[h], aq[h], bap != nullptr, a.bscale != nullptr, bsum, a.bact1, a.bact2, BP[h],
                              s1 + 8 * h, s2 + 8 * h);
            if (valid) *(uint4*)(y + pix * COUT + c0 + 8 * h) = o;
        }
    }
    float* prow = a.stats ? a.stats : (bsum ? a.bpart : nullptr);
    if (prow) {
        // sum over the 16 pixels of each lane row (DPP), then over the 4 waves; one partial row per workgroup
#pragma unroll
        for (int c = 0; c < CL; ++c) {
            float u = s1[c], q = s2[c];
            u += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, u), 0xB1, 0xF, 0xF, false));
            q += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, q), 0xB1, 0xF, 0xF, false));
            u += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, u), 0x4E, 0xF, 0xF, false));
            q += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, q), 0x4E, 0xF, 0xF, false));
            u += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, u), 0x141, 0xF, 0xF, false));
            q += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, q), 0x141, 0xF, 0xF, false));
            u += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, u), 0x140, 0xF, 0xF, false));
            q += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, q), 0x140, 0xF, 0xF, false));
            if (fr == 0) { sred[wid][0][c0 + c] = u; sred[wid][1][c0 + c] = q; }
        }
        __syncthreads();
        if (tid < COUT) {
            float u = 0.f, q = 0.f;
#pragma unroll
            for (int wv = 0; wv < 4; ++wv) { u += sred[wv][0][tid]; q += sred[wv][1][tid]; }
            float* dst = prow + (size_t)blockIdx.x * 2 * COUT;
            dst[tid] = u;
            // forward statistics: sum of squares; producer backward: sum du * xhat from sum du * z
            dst[COUT + tid] = a.stats ? q : a.brstd[tid] * (q - a.bmean[tid] * u);
        }
    }
}

int launch_pw(const GG& g, const FwdArgs& a, hipStream_t s) {
    const int blocks = pw_rows(g);
    const int64_t ngroups = ((int64_t)g.M + 15) / 16;
    const int gpw = (int)((ngroups + (int64_t)blocks * 4 - 1) / ((int64_t)blocks * 4));
    if (g.C1 == 64) PAI_LAUNCH((pw_k<64, 32>), dim3(blocks), dim3(256), 0, s, g, a, gpw);
    else PAI_LAUNCH((pw_k<32, 64>), dim3(blocks), dim3(256), 0, s, g, a, gpw);
    PAI_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------
// Weight gradient
// ------------------------------------------------------------------------------------
bool wgrad_mfma_ok(int dtype, const GG& g) {
    if (dtype != PAI_BF16) return false;
    if (g.lsw < 0) return false;   // the K loop addresses pixels with shifts: power-of-two image sizes only
    // pointwise convolutions of the finest attention gate (64 -> 32 and 32 -> 64 channels over 1 M pixels): the
    // 64 x 128 tile is a quarter full, but the launch is bound by reading the two operands once, not by the matrix pipe
    if (g.ntaps == 1 && g.C2 == 0 && ((g.C1 == 64 && g.Cout == 32) || (g.C1 == 32 && g.Cout == 64))) return true;
    // ... and the 64-input-channel pointwise convolutions of the residual blocks (half-full column tile)
    if (g.ntaps == 1 && g.C2 == 0 && g.C1 == 64 && (g.Cout % 64) == 0) return true;
    // ... and the 16- / 32-channel bottleneck convolutions (1x1 and 3x3) of the TransUNet / ResNet-50 encoder blocks:
    // both operands are staged in 8-channel chunks and every tile edge is guarded, so a partly empty tile is only
    // idle matrix rows in a launch that is bound by streaming the two activations once
    if (g.nphase == 1 && g.S == 1 && g.C2 == 0 && (g.C1 % 8) == 0 && (g.Cout % 8) == 0 && (g.C1 < 64 || g.Cout < 64))
        return true;
    if (g.C1 % 64 || g.C2 % 64) return false;
    if (g.Cout % 64) return false;
    // (the 128-column tile of the last column block may be partly empty -- 9 x 64 = 576 columns of a 64-channel 3x3
    //  convolution -- its loads and stores are guarded by xvalid / jcol)
    return true;
}

// XOR applied to the 16-B chunk index of row `row` in a 256-B-row LDS image: conflict-free for
// lane-linear row fills and for ds_read_b64_tr_b16 transposed reads (guide T10, layout (b))
__device__ __forceinline__ int tr_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ unsigned tr_off(int row, int ch) { return (unsigned)(256 * row + 16 * (ch ^ tr_swz(row))); }

template <int BMC>  // output-channel tile: 128 or 64
__global__ __launch_bounds__(256) void gg_wgrad_mfma_k(GG g, WgradArgs a, int cotiles, int jtiles,
                                                       int splits, int rows_per_split, int stage_lds) {
    constexpr int MT = BMC / 32;   // 16-row MFMA tiles per wave along co
    constexpr int BUF = 64 * 256;  // one staged operand tile: 64 pixel rows x 256 B
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Ys = smem;
    unsigned char* Xs = smem + BUF;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int jt = bid % jtiles; bid /= jtiles;
    const int cot = bid % cotiles; bid /= cotiles;
    const int split = bid % splits;
    const int ph = bid / splits;
    const int co0 = cot * BMC, j0 = jt * 128;
    // one writer per element of a pointwise layer's [Cout][Cin] gradient: the tile leaves through LDS as whole rows
    const bool stage_out = stage_lds != 0;

    const bf16_t* x1 = (const bf16_t*)a.x1;
    const bf16_t* x2 = (const bf16_t*)a.x2;
    const bf16_t* dy = (const bf16_t*)a.dy;
    const bf16_t* zero = (const bf16_t*)g_zero_line;

    // load map: position (row sr + 16j, slot sc) of the lane-linear image holds global chunk sc^swz(row)
    const int sc = lane & 15, sr = wid * 4 + (lane >> 4);
    const int gch = sc ^ tr_swz(sr);  // tr_swz(sr + 16j) == tr_swz(sr): only row bits 0..3 matter
    // gathered-column chunk -> (tap, source, channel)
    const int J = g.ntaps * g.Cin;
    const bool xvalid = j0 + gch * 8 < J;              // column tiles of a narrow problem are partly empty
    const int jc = xvalid ? j0 + gch * 8 : 0;
    const int xt = jc / g.Cin;
    const int xci = jc - xt * g.Cin;
    const int ddy = g.dy[ph][xt], ddx = g.dx[ph][xt];
    const bf16_t* xsrc;
    int xcs, xcc, xrelu;
    if (xci < g.C1) { xsrc = x1; xcs = g.C1; xcc = xci; xrelu = g.relu1; }
    else { xsrc = x2; xcs = g.C2; xcc = xci - g.C1; xrelu = g.relu2; }
    const bool yvalid = gch < BMC / 8 && (co0 + gch * 8) < g.Cout;
    // ReLU-on-load is per source tensor; a 128-wide column tile may straddle both sources
    const bool relu_any = g.relu1 || g.relu2;

    const int mbeg = split * rows_per_split;
    const int mend = min(g.M, mbeg + rows_per_split);
    const int niter = (mend - mbeg + 63) / 64;

    const int fi = lane & 15, fg = lane >> 4;
    const int tq = fi >> 2, tp = fi & 3;

    f4_t acc[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f4_t){0.f, 0.f, 0.f, 0.f};

    // which of this wave's four 16-column B tiles need ReLU (uniform per wave)
    bool nt_relu[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int jcol = j0 + wn * 64 + nt * 16;
        const int ci = jcol % g.Cin;
        nt_relu[nt] = jcol < J && (ci < g.C1 ? g.relu1 != 0 : g.relu2 != 0);
    }
    // prologue (pai_conv_wgrad_pro, pointwise layers): x is read as pact(x * pscale[c] + pshift[c]).  A B fragment is 8
    // pixels of ONE input channel per lane (column j0 + 64 wn + 16 nt + fi), so scale and shift are lane scalars.  Rows past
    // the end come from the zero line, turn into pact(pshift) and meet a zero dY row: no contribution.  (Requires finite
    // coefficients: with a non-finite scale / shift of a channel the out-of-range rows feed 0 x Inf = NaN into that
    // channel's dW column -- as the in-range rows of that channel do anyway, there and in the reference's own arithmetic,
    // so the result is the reference's NaN column, not a new failure)
    const bool pre = a.pscale != nullptr;
    float psc[4], psh[4];
    const float plo = a.pact == PAI_ACT_RELU ? 0.f : -INFINITY;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int jcol = j0 + wn * 64 + nt * 16 + fi;
        psc[nt] = (pre && jcol < J) ? a.pscale[jcol] : 0.f;
        psh[nt] = (pre && jcol < J) ? a.pshift[jcol] : 0.f;
    }

    // bias gradient: the workgroups of the first column tile (and first tap of each phase) see every
    // dY row of their channel tile exactly once -> column sums straight from the staged LDS tile
    // Block-diagonal (grouped) filter with 128 channels (pai_conv_desc.groups): a column tile is one tap x 128 input
    // channels and the wave (wm, wn) owns output channels 64 wm.. x input channels 64 wn.. -- off the diagonal the
    // filter is structurally zero and its gradient is not used, so those waves skip the matrix work (they still take
    // part in the fills and barriers and store zeros)
    const bool zero_block = BMC == 128 && g.gslice != 0 && g.Cin == 128 && g.Cout == 128 && wm != wn;
    const bool do_bias = a.dbias != nullptr && jt == 0;
    const int bc = tid % BMC, bh = tid / BMC;      // channel, row group (256/BMC groups)
    constexpr int BROWS = 64 / (256 / BMC);        // rows per group
    float bsum = 0.f;

    // power-of-two geometry: pixel <-> (n, y, x) by shifts (wgrad_mfma_ok guarantees it)
    const int los = g.OS == 2 ? 1 : 0, lss = g.S == 2 ? 1 : 0;
    const int poy = g.poy[ph], pox = g.pox[ph];
    const int ycol = co0 + gch * 8;
    // row pointers of the next 64-pixel step are computed while the current step's LDS-DMA is in flight
    const bf16_t* py[4];
    const bf16_t* px[4];
    auto prepare = [&](int it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = mbeg + it * 64 + sr + 16 * j;
            const bool mv = m < mend;
            const int gx = m & (g.OWg - 1);
            const int gy = (m >> g.lw) & (g.OHg - 1);
            const int n = m >> (g.lw + g.lh);
            const int opix = ((((n << g.ldh) + (gy << los) + poy) << g.ldw) + (gx << los) + pox);
            const bf16_t* p = dy + ((size_t)(unsigned)opix * (unsigned)g.Cout + ycol);
            py[j] = (mv && yvalid) ? p : zero;
            const int iy = (gy << lss) + ddy, ix = (gx << lss) + ddx;
            const bool inb = mv && xvalid && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
            const int spix = (((n << g.lsh) + iy) << g.lsw) + ix;
            const bf16_t* q = xsrc + ((size_t)(unsigned)spix * (unsigned)xcs + xcc);
            px[j] = inb ? q : zero;
        }
    };
    if (niter > 0) prepare(0);
    for (int it = 0; it < niter; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            GLDS16(py[j], Ys + (16 * j + wid * 4) * 256);
            GLDS16(px[j], Xs + (16 * j + wid * 4) * 256);
        }
        if (it + 1 < niter) prepare(it + 1);
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            if (zero_block) break;                    // grouped filter: this wave's 64 x 64 block lies off the diagonal
            bf8_t af[MT], bfr[4];
            const int row0 = kk * 32 + fg * 8 + tq;   // + 4 for the second half of the 8 k-values
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int ch = (wm * (BMC / 2) + mt * 16) / 8 + (tp >> 1);
                const bf4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                    (bf4_t __attribute__((address_space(3)))*)(Ys + tr_off(row0, ch) + 8 * (tp & 1)));
                const bf4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                    (bf4_t __attribute__((address_space(3)))*)(Ys + tr_off(row0 + 4, ch) + 8 * (tp & 1)));
                af[mt] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int ch = (wn * 64 + nt * 16) / 8 + (tp >> 1);
                const bf4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                    (bf4_t __attribute__((address_space(3)))*)(Xs + tr_off(row0, ch) + 8 * (tp & 1)));
                const bf4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                    (bf4_t __attribute__((address_space(3)))*)(Xs + tr_off(row0 + 4, ch) + 8 * (tp & 1)));
                bfr[nt] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
            if (relu_any) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    if (nt_relu[nt]) bfr[nt] = relu_frag(bfr[nt]);
            }
            if (pre) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const uint4 u = __builtin_bit_cast(uint4, bfr[nt]);
                    const unsigned wv[4] = {u.x, u.y, u.z, u.w};
                    unsigned o[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float lo = fmaxf(fmaf(__uint_as_float(wv[i] << 16), psc[nt], psh[nt]), plo);
                        const float hi = fmaxf(fmaf(__uint_as_float(wv[i] & 0xffff0000u), psc[nt], psh[nt]), plo);
                        o[i] = pk2bf(lo, hi);
                    }
                    bfr[nt] = __builtin_bit_cast(bf8_t, make_uint4(o[0], o[1], o[2], o[3]));
                }
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt], bfr[nt], acc[mt][nt], 0, 0, 0);
        }
        if (do_bias) {
#pragma unroll 8
            for (int r = 0; r < BROWS; ++r) {
                const int row = bh * BROWS + r;
                bsum += bf2f(*(const bf16_t*)(Ys + tr_off(row, bc >> 3) + (bc & 7) * 2));
            }
        }
        __syncthreads();
    }
    (void)xrelu;
    if (do_bias) {
        float* red = (float*)smem;   // all tile reads are behind the loop's final barrier
        red[tid] = bsum;
        __syncthreads();
        if (tid < BMC && co0 + tid < g.Cout) {
            float t = 0.f;
#pragma unroll
            for (int h = 0; h < 256 / BMC; ++h) t += red[tid + h * BMC];
            if (splits == 1 && g.nphase == 1) {
                if (a.overwrite_bias) a.dbias[co0 + tid] = t;
                else a.dbias[co0 + tid] += t;
            } else {
                atomicAdd(a.dbias + co0 + tid, t);
            }
        }
    }

    // ---- un-split pointwise layers (nn.Linear of the ViT bottleneck: 128 token rows, 4096 x 4096 .. 12288 weights; the
    // launch is the store of dW): the tile goes through LDS in two 64-row halves and leaves as 512-byte rows -- 32 lanes x
    // 16 bytes per row instead of 16 scalar stores of 64 bytes per MFMA tile (28 -> ~14 us at 4096 x 4096, 2.4 -> ~5 TB/s)
    if (stage_out) {
        constexpr int SP = 132;                                // padded row: 128 columns + 4
        float* st = (float*)smem;                              // 64 x 132 x 4 B = 33 KB (the launch asks for it)
        constexpr int HALVES = BMC / 64;
#pragma unroll
        for (int hpass = 0; hpass < HALVES; ++hpass) {
            __syncthreads();                                   // the K loop / the previous half is done with smem
            if (HALVES == 1 || wm == hpass) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = (HALVES == 1 ? wm * (BMC / 2) : 0) + mt * 16 + fg * 4 + r;
                            st[row * SP + wn * 64 + nt * 16 + fi] = acc[mt][nt][r];
                        }
            }
            __syncthreads();
            const int c4 = (tid & 31) * 4;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = (tid >> 5) + 8 * i;
                const int co = co0 + hpass * 64 + row, jcol = j0 + c4;
                if (co < g.Cout && jcol < g.Cin) {
                    const float4 v = *(const float4*)(st + row * SP + c4);
                    float* pw = a.dw + (size_t)co * g.Cin + jcol;
                    if (a.overwrite) *(float4*)pw = v;
                    else {
                        float4 o = *(float4*)pw;
                        o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
                        *(float4*)pw = o;
                    }
                }
            }
        }
        return;
    }

    // ---- accumulate into the fp32 gradient (fwd pack) ------------------------------
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int jcol = j0 + wn * 64 + nt * 16 + fi;
        if (jcol >= g.ntaps * g.Cin) continue;
        const int t = jcol / g.Cin;
        const int ci = jcol - t * g.Cin;
        const size_t cbase = (size_t)g.wt[ph][t] * g.Cin + ci;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + wm * (BMC / 2) + mt * 16 + fg * 4 + r;
                if (co < g.Cout) {
                    float* pw = a.dw + (size_t)co * g.wtaps * g.Cin + cbase;
                    // un-split: this workgroup is the only writer of the element (taps of different phases are
                    // disjoint) -> plain read-modify-write at store bandwidth instead of the ~1.3 TB/s atomic rate
                    if (splits == 1) {
                        if (a.overwrite) *pw = acc[mt][nt][r];     // no zero-fill pass, no read of dW (skinny ViT layers)
                        else *pw += acc[mt][nt][r];
                    } else {
                        atomicAdd(pw, acc[mt][nt][r]);
                    }
                }
            }
        }
    }
}


// Patch-resident weight gradient.  Same LDS arithmetic as the forward kernel: per 64-pixel step the kernel
// above fills 32 KB (dY tile + gathered X tile) for 512 MFMA cycles.  Here the column tile is one 2 x 2
// tap window x 32 input channels, and the K step is a 4 x 16 block of output pixels: the four taps read
// the same 5 x 17 source pixels (64 B each), so the X fill drops from 16 KB to 5.4 KB per step.
//   X patch image: pixel p = py * 17 + px at byte 64 p; its two 32-B halves are swapped when bit 3 of p is
//   set, which keeps the two 8-row groups of a ds_read_b64_tr_b16 on different banks for every tap shift.
// (measured and dropped: two LDS stages with a counted vmcnt + raw barrier, 48 KB per workgroup, still three per CU:
//  10-50 % SLOWER -- decoders[6] 194 -> 306 us, decoders[5] 145 -> 169 us; this kernel's latency is hidden across
//  workgroups, and the second stage only adds LDS-DMA pressure)
// Compile-time timing ablations of gg_wgrad_patch_k (results WRONG; scripts/micro/variants.sh, scripts/micro/wg_abl.sh):
// 1 no dW accumulation (atomics / stores), 2 no MFMA, 4 no LDS-DMA fills, 8 no fragment reads.  Measured (round 2, one box,
// weight gradient of decoders[4] / D block 3 / decoders[6], us): everything 156 / 150 / 184; no dW accumulation
// 127 / 127 / 177; no fills 116 / 117 / 127; no fragment reads 154 / 148 / 182; neither fills nor reads 117 / 115 / 126;
// MFMAs and loop alone 104 / 105 / 124 -- the transposed fragment reads are hidden, the fp32 atomics of the eight pixel
// splits cost 19 %, the fills (10 B per kFLOP, 1.7 x the forward kernel's) 26 %, and the bare loop still runs a third
// behind the forward kernel's (two vector instructions per MFMA: 32 of the 66 are the 8-byte transposed reads).
#ifndef WGRAD_ABL
#define WGRAD_ABL 0
#endif
template <int BMC>
__global__ __launch_bounds__(256, 4) void gg_wgrad_patch_k(GG g, WgradArgs a, PatchGeo pg, int cotiles, int jtiles,
                                                        int splits, int blocks_per_split) {
    constexpr int MT = BMC / 32;
    constexpr int YBUF = 64 * 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Ys = smem;
    unsigned char* Xs = smem + YBUF;   // 128 pixels x 64 B (85 used)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int jt = bid % jtiles; bid /= jtiles;
    const int cot = bid % cotiles; bid /= cotiles;
    const int split = bid % splits;
    const int ph = bid / splits;
    const int co0 = cot * BMC;
    const int q = jt & (pg.groups - 1);
    const int ci0 = (jt >> (pg.groups == 4 ? 2 : 0)) * 32;

    const bf16_t* dy = (const bf16_t*)a.dy;
    const bf16_t* zero = (const bf16_t*)g_zero_line;
    const bool second = ci0 >= g.C1;
    const bf16_t* xsrc = second ? (const bf16_t*)a.x2 : (const bf16_t*)a.x1;
    const int xcs = second ? g.C2 : g.C1;
    const int xrelu = second ? g.relu2 : g.relu1;

    // Both tiles are filled by LDS-DMA through buffer descriptors (buffer_load_dwordx4 ... lds): the per-lane part of
    // an address is a 32-bit byte offset, the per-step part rides in the instruction's scalar offset, and a lane whose
    // offset lies beyond the buffer writes ZEROS to LDS (scripts/micro/oob_probe.hip) -- padding pixels, pixels beyond
    // the patch and absent channels need no zero line and no select.  What is left per step and thread: one compare
    // pair + select per X piece.  (The pointer form cost ~70 vector instructions per 32 MFMAs; the SIMD issues two
    // per MFMA at most before the matrix pipe starves.)
    // host: every tensor of the layer is smaller than 2 GB, so vector + scalar offset of an absent lane cannot wrap
    // back into the buffer whether or not the range check counts the scalar part
    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(a.dy), 0, (unsigned)((g.N << (g.ldh + g.ldw)) * g.Cout) * 2u, 0x00020000);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>((const void*)xsrc), 0, (unsigned)(g.N * g.H * g.W * xcs) * 2u, 0x00020000);
    // dY tile fill map (as gg_wgrad_mfma_k): row sr + 16 j = pixel (gy0 + j, gx0 + sr) of the step's block
    const int sc = lane & 15, sr = wid * 4 + (lane >> 4);
    const int gch = sc ^ tr_swz(sr);
    const bool yvalid = gch < BMC / 8 && (co0 + gch * 8) < g.Cout;
    const int los = g.OS == 2 ? 1 : 0;
    const int poy = g.poy[ph], pox = g.pox[ph];
    // byte offset of this thread's chunk from the step's first output pixel; beyond the buffer for absent channels
    const unsigned ythr = yvalid ? (unsigned)((sr << los) * g.Cout + co0 + gch * 8) * 2u : OOB;
    const unsigned yrow1 = (unsigned)(((1 << los) << g.ldw) * g.Cout) * 2u;    // bytes between the thread's four rows
    // X patch fill map: thread -> (pixel 64 jj + tid / 4, 16-B slot tid % 4).  Invariant per thread: the pixel's
    // source coordinates relative to the step's origin (beyond-the-patch pixels get a row no image has) and the
    // byte offset of its chunk from the origin pixel.
    const int xs = tid & 3;
    const int wby = pg.by[ph][q], wbx = pg.bx[ph][q];
    int xty[2], xtx[2];
    unsigned xthr[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int p = jj * 64 + (tid >> 2);
        const int py_ = p / PATCH_W, px_ = p - py_ * PATCH_W;
        xty[jj] = p >= 5 * PATCH_W ? 0x40000000 : py_ * g.S + wby;
        xtx[jj] = px_ * g.S + wbx;
        const int xch = (second ? ci0 - g.C1 : ci0) + ((xs ^ (((p >> 3) & 1) << 1)) * 8);
        xthr[jj] = (unsigned)(((py_ * g.S + wby) * g.W + px_ * g.S + wbx) * xcs + xch) * 2u;
    }

    const int lbx = g.lw - 4, lby = g.lh - 2;
    const int kb0 = split * blocks_per_split;
    const int kb1 = min(g.M >> 6, kb0 + blocks_per_split);

    const int fi = lane & 15, fg = lane >> 4;
    const int tq = fi >> 2, tp = fi & 3;
    // X fragment addresses: row r = kk * 32 + fg * 8 + tq (+ 4) of the step -> patch pixel (r >> 4) * 17 + (r & 15)
    unsigned xrow[2][2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = kk * 32 + fg * 8 + tq + 4 * h;
            xrow[kk][h] = (unsigned)((r >> 4) * PATCH_W + (r & 15));
        }
    const unsigned toff4 = pg.toff4[ph][q];
    // Fragment addresses: one base per k-half for dY, one per (kk, tap, k-half) for X, plus an XOR constant per 16-column
    // tile (the swizzles only touch address bits the tile index owns alone).  hipcc otherwise keeps all 32 addresses in
    // registers across the K loop: 162 VGPRs = three workgroups per CU; with 128 a fourth one fits, and this kernel
    // hides its LDS-DMA latency across workgroups (scripts/micro/convbench: fills alone and MFMAs alone take ~110 us each
    // on decoders[4], together 150-160).
    unsigned ybase[2], xbase[2][2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int rowl = fg * 8 + tq + 4 * h;
        ybase[h] = (unsigned)(256 * rowl + 16 * ((wm * (BMC / 16) + (tp >> 1)) ^ tr_swz(rowl)) + 8 * (tp & 1));
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                const unsigned p = xrow[kk][h] + ((toff4 >> (8 * (wn * 2 + t2))) & 0xffu);
                xbase[kk][t2][h] = (unsigned)YBUF + p * 64 + (((p >> 3) & 1u) << 5) + tp * 8;
            }
    }
#define WGP_XOR(dst, src, imm) asm volatile("v_xor_b32 %0, %2, %1" : "=v"(dst) : "v"(src), "s"(imm))
    // Fragment reads take the LDS byte address as an integer: the dynamic LDS block of this kernel (it has no static
    // one) starts at LDS address 0, checked here.  `smem + offset` costs a v_add_u32 with the link-time constant 0 per
    // read, 44 per K-step.
#define WGP_TR(addr) __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf4_t __attribute__((address_space(3)))*)(size_t)(unsigned)(addr))
    if ((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem != 0u) __builtin_trap();

    f4_t acc[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f4_t){0.f, 0.f, 0.f, 0.f};

    const bool do_bias = a.dbias != nullptr && jt == 0;
    const int bc = tid % BMC, bh = tid / BMC;
    constexpr int BROWS = 64 / (256 / BMC);
    float bsum = 0.f;

    // step kb: scalar origin of its output block / source patch (byte offsets), then the per-thread X offsets
    unsigned ysof = 0, xofs[2] = {0u, 0u};
    auto prepare = [&](int kb) {
        const int gx0 = (kb & ((1 << lbx) - 1)) << 4;
        const int gy0 = ((kb >> lbx) & ((1 << lby) - 1)) << 2;
        const int n = kb >> (lbx + lby);
        ysof = (unsigned)(((((n << g.ldh) + (gy0 << los) + poy) << g.ldw) + (gx0 << los) + pox) * g.Cout) * 2u;
        const int oy = gy0 * g.S, ox = gx0 * g.S;
        const unsigned xsof = (unsigned)(((((n << g.lsh) + oy) << g.lsw) + ox) * xcs) * 2u;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const bool inb = (unsigned)(xty[jj] + oy) < (unsigned)g.H && (unsigned)(xtx[jj] + ox) < (unsigned)g.W;
            xofs[jj] = inb ? xthr[jj] + xsof : OOB;
        }
    };
#define WGP_BLDS16(rs, voff, soff, lptr) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lptr), 16, (int)(voff), (int)(soff), 0, 0)
    if (kb0 < kb1) prepare(kb0);
    for (int kb = kb0; kb < kb1; ++kb) {
        if (!(WGRAD_ABL & 4)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) WGP_BLDS16(yrs, ythr, ysof + (unsigned)j * yrow1, Ys + (16 * j + wid * 4) * 256);
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) WGP_BLDS16(xrs, xofs[jj], 0, Xs + (jj * 64 + wid * 16) * 64);
        }
        if (kb + 1 < kb1) prepare(kb + 1);
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf8_t af[MT], bfr[4];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                // channel chunk (wm * BMC/16 + 2 mt + (tp >> 1)) ^ swz(row): the tile index is an XOR of address bits 5-6
                unsigned a0 = ybase[0], a1 = ybase[1];
                if (mt) { WGP_XOR(a0, ybase[0], mt << 5); WGP_XOR(a1, ybase[1], mt << 5); }
                const bf4_t lo = WGP_TR(a0 + kk * 8192);
                const bf4_t hi = WGP_TR(a1 + kk * 8192);
                af[mt] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                unsigned o0 = xbase[kk][nt >> 1][0], o1 = xbase[kk][nt >> 1][1];
                if (nt & 1) { WGP_XOR(o0, xbase[kk][nt >> 1][0], 32); WGP_XOR(o1, xbase[kk][nt >> 1][1], 32); }
                const bf4_t lo = WGP_TR(o0);
                const bf4_t hi = WGP_TR(o1);
                bfr[nt] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
            if (xrelu) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) bfr[nt] = relu_frag(bfr[nt]);
            }
            if (WGRAD_ABL & 2) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt][0][0] += (float)af[mt][0] + (float)bfr[mt & 3][0];
            } else {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt], bfr[nt], acc[mt][nt], 0, 0, 0);
            }
        }
        if (do_bias) {
#pragma unroll 8
            for (int r = 0; r < BROWS; ++r) {
                const int row = bh * BROWS + r;
                bsum += bf2f(*(const bf16_t*)(Ys + tr_off(row, bc >> 3) + (bc & 7) * 2));
            }
        }
        __syncthreads();
    }
    if (do_bias) {
        float* red = (float*)smem;
        red[tid] = bsum;
        __syncthreads();
        if (tid < BMC && co0 + tid < g.Cout) {
            float t = 0.f;
#pragma unroll
            for (int h = 0; h < 256 / BMC; ++h) t += red[tid + h * BMC];
            if (splits == 1 && g.nphase == 1) {
                if (a.overwrite_bias) a.dbias[co0 + tid] = t;
                else a.dbias[co0 + tid] += t;
            } else {
                atomicAdd(a.dbias + co0 + tid, t);
            }
        }
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int wt = (int)((pg.wt4[ph][q] >> (8 * (wn * 2 + (nt >> 1)))) & 0xffu);
        const size_t cbase = (size_t)wt * g.Cin + ci0 + (nt & 1) * 16 + fi;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + wm * (BMC / 2) + mt * 16 + fg * 4 + r;
                if (co < g.Cout) {
                    float* pw = a.dw + (size_t)co * g.wtaps * g.Cin + cbase;
                    // un-split: this workgroup is the only writer of the element (taps of different phases are
                    // disjoint) -> plain read-modify-write at store bandwidth instead of the ~1.3 TB/s atomic rate
                    if (WGRAD_ABL & 1) { if (acc[mt][nt][r] == 123.456f) *pw = 0.f; }
                    else if (splits == 1) {
                        if (a.overwrite) *pw = acc[mt][nt][r];     // no zero-fill pass, no read of dW (skinny ViT layers)
                        else *pw += acc[mt][nt][r];
                    } else {
                        atomicAdd(pw, acc[mt][nt][r]);
                    }
                }
            }
        }
    }
}

int launch_colsum(int dtype, const void* x, int64_t rows, int C, float* out, hipStream_t s);

// shapes of the patch-resident weight gradient: full 64-channel output tiles only (the partly filled tiles of the
// small-channel layers go to gg_wgrad_mfma_k, whose tile edges are guarded)
static bool wgrad_patch_shape_ok(const GG& g) {
    // 32-bit byte offsets into buffer descriptors: every tensor below 2 GB
    const int64_t xb = (int64_t)g.N * g.H * g.W * (g.C1 > g.C2 ? g.C1 : g.C2) * 2, yb = (int64_t)g.N * g.OH * g.OW * g.Cout * 2;
    if (xb >= (1ll << 31) || yb >= (1ll << 31)) return false;
    return g.lw >= 4 && g.lh >= 2 && (g.C1 % 32) == 0 && (g.C2 % 32) == 0 && (g.Cout % 64) == 0 && g.Cin >= 64;
}

static int wgrad_mfma_splits(const GG& g, int* rows_out);
static bool wgrad_mfma_uses_patch(const GG& g) {
    static const bool no_patch = getenv("PAI_NO_WPATCH") && atoi(getenv("PAI_NO_WPATCH")) != 0;
    PatchGeo pg;
    return !no_patch && wgrad_patch_shape_ok(g) && patch_geo(g, 4, &pg);
}
bool wgrad_mfma_can_overwrite(const GG& g) {
    int rows;
    if (wgrad3_ok(g)) return wgrad3_overwrites(g);
    // un-split gg_wgrad_mfma_k: one writer per dW element (the taps of different phases are disjoint); the bias sums of
    // several phases meet by atomics, launch_wgrad_mfma clears dbias for them
    return !wgrad_mfma_uses_patch(g) && wgrad_mfma_splits(g, &rows) == 1;
}

// 128-channel tiles unless the layer would then run as at most one un-split workgroup per CU (the bottleneck layers:
// <= 512 tiles, <= 2048 pixels): every K step of a lone workgroup is an exposed fill -> barrier -> multiply round trip, and
// 64-channel tiles put two or more workgroups on a CU (tunable wgrad_narrow; convbench, us: encoders[5] 51.9 -> 37.5, [6] 24.7 ->
// 18.4, [7] 19.2 -> 13.8, decoders[0] 19.5 -> 14.5, [1] 35.1 -> 31.3, [2] 63.5 -> 62.5; bit-identical on integer data)
static bool wgrad_mfma_big(const GG& g) {
    if ((g.Cout % 128) != 0) return false;
    const int narrow = pai_tunable("wgrad_narrow", 512);      // 64-channel tiles up to this many 128-channel tiles (0: never)
    if (narrow && !wgrad_mfma_uses_patch(g)) {
        const int tiles128 = (g.Cout / 128) * cdiv(g.ntaps * g.Cin, 128) * g.nphase;
        if (tiles128 <= narrow && g.M <= 2048) return false;
    }
    return true;
}

static int wgrad_mfma_splits(const GG& g, int* rows_out) {
    const bool big = wgrad_mfma_big(g);
    const int cotiles = big ? g.Cout / 128 : cdiv(g.Cout, 64);
    const int jtiles = cdiv(g.ntaps * g.Cin, 128);
    const int tiles = cotiles * jtiles * g.nphase;
    // Split of the pixel range: enough workgroups to fill the chip (3 per CU for the 128-wide tile, 4 for the
    // lighter 64-wide one), but every split adds one fp32 atomic pass over dW -- for the small-image layers
    // that pass, not the MFMA loop, is the cost, so a split never gets fewer than 512 pixels.
    static const int target_env = getenv("PAI_WGRAD_TARGET") ? atoi(getenv("PAI_WGRAD_TARGET")) : 0;
    const int target_tun = pai_tunable("wgrad_target", 0);
    // gg_wgrad_patch_k fits four workgroups per CU (128 VGPRs), gg_wgrad_mfma_k<128> three.  Measured per layer
    // (scripts/micro/convbench --set wgrad_target=...): the 137-GFLOP layers (decoders[4-6], D blocks 1-3 at 2N) want
    // four rounds' worth of workgroups (1024: 138-165 us against 145-224 at 512), the 69-GFLOP ones (encoders[1-3]) and
    // everything on gg_wgrad_mfma_k two (512: 86-88 us against 95-99 at 1024; encoders[4] 62 against 67 at 768) -- there
    // the extra atomic passes over dW cost more than the better balance buys.
    const double gflop = 2.0 * (double)g.M * g.nphase * g.Cout * g.ntaps * g.Cin * 1e-9;
    const int target_def = (wgrad_mfma_uses_patch(g) && gflop >= 100.0) ? 1024 : 512;
    const int target = target_tun ? target_tun : (target_env ? target_env : target_def);
    int splits = cdiv(target, tiles);
    static const int min_rows = getenv("PAI_WGRAD_MINROWS") ? atoi(getenv("PAI_WGRAD_MINROWS")) : 512;
    const int max_splits = cdiv(g.M, min_rows);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    // An un-split launch owns every dW element in exactly one workgroup and updates it with a plain
    // read-modify-write; a split one adds splits x |dW| bytes of float atomics (~1.3 TB/s chip-wide).  For the
    // bottleneck layers (<= 2048 pixels, 4-8 M weights) those atomics were the whole cost.
    static const int unsplit_rows = getenv("PAI_WGRAD_UNSPLIT_ROWS") ? atoi(getenv("PAI_WGRAD_UNSPLIT_ROWS")) : 2048;
    if ((tiles >= 256 && g.M <= unsplit_rows) || (tiles >= 512 && g.M <= 2 * unsplit_rows)) splits = 1;
    int rows = cdiv(cdiv(g.M, splits), 64) * 64;
    splits = cdiv(g.M, rows);
    *rows_out = rows;
    return splits;
}

// pointwise layers whose weight gradient can read x through a prologue (gg_wgrad_mfma_k only)
bool wgrad_pro_ok(int dtype, const GG& g) {
    return wgrad_mfma_ok(dtype, g) && g.ntaps == 1 && g.nphase == 1 && g.C2 == 0 && !g.gslice && !g.relu1 && !wgrad3_ok(g);
}

int launch_wgrad_mfma(const GG& g, const WgradArgs& a, hipStream_t s) {
    if (a.pscale && !wgrad_pro_ok(PAI_BF16, g)) {
        pai_set_error("launch_wgrad_mfma: this layer's weight gradient takes no prologue");
        return 1;
    }
    if (wgrad3_ok(g)) return launch_wgrad3(g, a, s);
    const bool big = wgrad_mfma_big(g);
    const int cotiles = big ? g.Cout / 128 : cdiv(g.Cout, 64);
    const int jtiles = cdiv(g.ntaps * g.Cin, 128);
    const int tiles = cotiles * jtiles * g.nphase;
    int rows;
    const int splits = wgrad_mfma_splits(g, &rows);
    static const bool no_patch = getenv("PAI_NO_WPATCH") && atoi(getenv("PAI_NO_WPATCH")) != 0;
    PatchGeo pg;
    if (!a.pscale && !no_patch && wgrad_patch_shape_ok(g) && patch_geo(g, 4, &pg)) {
        const int kblocks = g.M / 64;
        const int per = rows / 64;
        const int psplits = cdiv(kblocks, per);
        const size_t plds = 64 * 256 + 128 * 64;
        dim3 pgrid(tiles * psplits);
        if (big)
            PAI_LAUNCH(gg_wgrad_patch_k<128>, pgrid, dim3(256), plds, s, g, a, pg, cotiles, jtiles, psplits, per);
        else
            PAI_LAUNCH(gg_wgrad_patch_k<64>, pgrid, dim3(256), plds, s, g, a, pg, cotiles, jtiles, psplits, per);
        PAI_LAUNCH_CHECK();
        return 0;
    }
    // un-split pointwise layers store their tile as whole rows through LDS (see the kernel)
    const int stage = splits == 1 && g.ntaps == 1 && g.nphase == 1 && (g.Cin % 4) == 0 && !g.gslice &&
                      pai_tunable("wgrad_stage", 1) != 0;
    const size_t lds = stage ? 64 * 132 * 4 : 2 * 64 * 256;
    dim3 grid(tiles * splits);
    if (a.overwrite_bias && a.dbias && g.nphase > 1) {
        hipError_t e = pai::memset_async(a.dbias, 0, (size_t)g.Cout * sizeof(float), s);
        PAI_CHECK(e == hipSuccess, "launch_wgrad_mfma: hipMemsetAsync: %s", hipGetErrorString(e));
    }
    if (big)
        PAI_LAUNCH(gg_wgrad_mfma_k<128>, grid, dim3(256), lds, s, g, a, cotiles, jtiles, splits, rows, stage);
    else
        PAI_LAUNCH(gg_wgrad_mfma_k<64>, grid, dim3(256), lds, s, g, a, cotiles, jtiles, splits, rows, stage);
    PAI_LAUNCH_CHECK();
    return 0;
}

const char* wgrad_mfma_kernel_name(const GG& g) {
    if (wgrad3_ok(g)) return wgrad3_kernel_name(g);
    const bool big = wgrad_mfma_big(g);
    const bool no_patch = getenv("PAI_NO_WPATCH") && atoi(getenv("PAI_NO_WPATCH")) != 0;
    PatchGeo pg;
    if (!no_patch && wgrad_patch_shape_ok(g) && patch_geo(g, 4, &pg))
        return big ? "gg_wgrad_patch_k<128>" : "gg_wgrad_patch_k<64>";
    return big ? "gg_wgrad_mfma_k<128>" : "gg_wgrad_mfma_k<64>";
}

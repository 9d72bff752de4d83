// Patch-resident weight gradient with a 128 x 256 output tile ("wgrad2"): dW[co][tap][ci] += sum_pixels dY[pix][co] * X[pix + tap][ci].
//
// gg_wgrad_patch_k (gg_mfma.hip) computes a 128 (output channels) x 128 (2 x 2 taps x 32 input channels) tile per
// workgroup and fills 16 KB of dY + 5.4 KB of X per 64-pixel step for 512 matrix cycles per SIMD: 42 B of LDS-DMA per
// matrix cycle and CU, more than the ~70 GB/s per CU the L2 -> LDS path delivers, so the kernel is fill-bound
// (700-930 TFLOP/s isolated against 1100-1190 for the forward kernel, whose 256 x 128 tile needs 26).
// Here the column tile is 2 x 2 taps x 64 input channels = 256 columns, eight waves (2 over the output channels x 4
// taps) of 64 x 64: the dY tile is shared by twice the columns and the step fills 16 KB + 10.9 KB for 1024 matrix
// cycles = 26 B per cycle, the forward kernel's ratio.
//   X patch image: 5 x 17 source pixels x 64 channels, pixel p = py * 17 + px at byte 128 p; its four 32-B segments
//   (16 channels each) are stored at segment s ^ f(p), f(p) = bit 1 of p | bit 3 of p << 1: the eight pixel rows a
//   half-wave of a ds_read_b64_tr_b16 touches (p0 .. p0+3 and p0+8 .. p0+11, any p0 -- the tap shift) then cover the 64
//   banks once (checked exhaustively, scripts/lds_swizzle_check.py).
//
// STATUS (round 2, scripts/micro/convbench on one MI355X, bit-exact against gg_wgrad_patch_k on integer data): NOT the
// default (tunable "wgrad2").  decoders[4] weight gradient, 137 GFLOP: gg_wgrad_patch_k 150-160 us, this kernel 182 us
// at 512 workgroups (246 us at 640: one and a quarter rounds of two workgroups per CU).  Second attempt with everything
// that helped gg_wgrad_patch_k (LDS-DMA through buffer descriptors, integer LDS addresses) plus two LDS stages (the
// next step's tiles in flight during the multiplications, one barrier per step): 156 us against 146 for
// gg_wgrad_patch_k on the same box (decoders[5] 158 / 139, D block 2 151 / 136, encoders[2] 99 / 95).  Ablations of gg_wgrad_patch_k
// (WGRAD_ABL): MFMA + fragment reads alone 108 us, LDS-DMA fills alone 109 us, no atomics 150 us, and a fourth workgroup
// per CU buys 3 %: the two halves do not overlap because both live on the LDS port (64 KB of transposed fragment
// reads + 21 KB of DMA writes per 512 matrix cycles), not because of L2 -> LDS latency or fill volume -- halving the
// fill per FLOP, as this kernel does, is not what the weight gradient needs.
//
// Serves the weight-gradient half of aten::convolution_backward of the dense Conv2d / ConvTranspose2d k4 s2 p1 layers
// (models/pix2pix.py:58-111, models/wrapper.py:229-232), like gg_wgrad_patch_k.
#include "gg_tile.h"

__device__ __forceinline__ int tr_swz2(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ unsigned tr_off2(int row, int ch) { return (unsigned)(256 * row + 16 * (ch ^ tr_swz2(row))); }
__device__ __forceinline__ unsigned xseg_swz(unsigned p) { return ((p >> 1) & 1u) | (((p >> 3) & 1u) << 1); }

template <int BMC>   // output-channel tile (128)
__global__ __launch_bounds__(512, 4) void gg_wgrad_patch2_k(GG g, WgradArgs a, PatchGeo pg, int cotiles, int jtiles,
                                                            int splits, int blocks_per_split) {
    constexpr int MT = BMC / 32;
    constexpr int YBUF = 64 * 256;
    constexpr int STAGE = YBUF + 128 * 128;      // dY tile + X patch (128 pixel slots x 128 B, 85 used); two stages
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;       // output-channel half, tap of the window
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int jt = bid % jtiles; bid /= jtiles;
    const int cot = bid % cotiles; bid /= cotiles;
    const int split = bid % splits;
    const int ph = bid / splits;
    const int co0 = cot * BMC;
    const int q = jt & (pg.groups - 1);
    const int ci0 = (jt >> (pg.groups == 4 ? 2 : 0)) * 64;

    const bool second = ci0 >= g.C1;
    const bf16_t* xsrc = second ? (const bf16_t*)a.x2 : (const bf16_t*)a.x1;
    const int xcs = second ? g.C2 : g.C1;
    const int xrelu = second ? g.relu2 : g.relu1;
    // LDS-DMA through buffer descriptors, as in gg_wgrad_patch_k: 32-bit byte offsets, the per-step part in the scalar
    // offset, out-of-range lanes write zeros (host: every tensor below 2 GB)
    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(a.dy), 0, (unsigned)((g.N << (g.ldh + g.ldw)) * g.Cout) * 2u, 0x00020000);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>((const void*)xsrc), 0, (unsigned)(g.N * g.H * g.W * xcs) * 2u, 0x00020000);

    // dY tile fill map: row r = sr + 32 j = pixel (gy0 + (r >> 4), gx0 + (r & 15)) of the step's 4 x 16 block
    const int sc = lane & 15, sr = wid * 4 + (lane >> 4);
    const int gch = sc ^ tr_swz2(sr);                  // tr_swz2 only looks at row bits 0..3: the same for r + 32
    const bool yvalid = gch < BMC / 8 && (co0 + gch * 8) < g.Cout;
    const int los = g.OS == 2 ? 1 : 0;
    const int poy = g.poy[ph], pox = g.pox[ph];
    const unsigned ythr = yvalid ? (unsigned)(((((sr >> 4) << los) << g.ldw) + ((sr & 15) << los)) * g.Cout + co0 + gch * 8) * 2u : OOB;
    const unsigned yrow2 = (unsigned)(((2 << los) << g.ldw) * g.Cout) * 2u;   // rows r and r + 32: two pixel rows apart
    // X patch fill map: thread -> (pixel 64 jj + tid / 8, 16-B chunk tid % 8)
    const int wby = pg.by[ph][q], wbx = pg.bx[ph][q];
    int xty[2], xtx[2];
    unsigned xthr[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int p = jj * 64 + (tid >> 3);
        const int py_ = p / PATCH_W, px_ = p - py_ * PATCH_W;
        xty[jj] = p >= 5 * PATCH_W ? 0x40000000 : py_ * g.S + wby;      // beyond the patch: a row no image has
        xtx[jj] = px_ * g.S + wbx;
        const int c = tid & 7;                   // physical chunk: segment c >> 1 holds source segment (c >> 1) ^ f(p)
        const int xch = (second ? ci0 - g.C1 : ci0) + (((((c >> 1) ^ (int)xseg_swz((unsigned)p)) << 1) | (c & 1)) * 8);
        xthr[jj] = (unsigned)(((py_ * g.S + wby) * g.W + px_ * g.S + wbx) * xcs + xch) * 2u;
    }

    const int lbx = g.lw - 4, lby = g.lh - 2;
    const int kb0 = split * blocks_per_split;
    const int kb1 = min(g.M >> 6, kb0 + blocks_per_split);

    const int fi = lane & 15, fg = lane >> 4;
    const int tq = fi >> 2, tp = fi & 3;
    // X fragment rows: K index r = kk * 32 + fg * 8 + tq (+ 4) of the step -> patch pixel (r >> 4) * 17 + (r & 15) + tap shift
    const unsigned toff = (pg.toff4[ph][q] >> (8 * wn)) & 0xffu;
    unsigned xrow[2][2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = kk * 32 + fg * 8 + tq + 4 * h;
            xrow[kk][h] = (unsigned)((r >> 4) * PATCH_W + (r & 15)) + toff;
        }

    // Fragment addresses as ONE base per (k-half h) for dY and per (kk, h) for X plus an XOR constant per 16-column
    // tile: the swizzles only touch address bits 5-6 (dY: bits 4-7), which the tile index owns alone.  hipcc otherwise
    // keeps all 32 addresses in registers across the K loop (161 VGPRs: one workgroup per CU instead of two).
    unsigned ybase[2], xbase[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int rowl = fg * 8 + tq + 4 * h;                                  // row within a 32-pixel half step
        ybase[h] = (unsigned)(256 * rowl + 16 * ((wm * (BMC / 16) + (tp >> 1)) ^ tr_swz2(rowl)) + 8 * (tp & 1));
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
            xbase[kk][h] = (unsigned)YBUF + xrow[kk][h] * 128 + (xseg_swz(xrow[kk][h]) << 5) + tp * 8;
    }
#define WG2_XOR(dst, src, imm) asm volatile("v_xor_b32 %0, %2, %1" : "=v"(dst) : "v"(src), "s"(imm))
    // integer LDS addresses (the dynamic LDS block starts at 0: checked), see gg_wgrad_patch_k
#define WG2_TR(addr) __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf4_t __attribute__((address_space(3)))*)(size_t)(unsigned)(addr))
#define WG2_BLDS16(rs, voff, soff, laddr) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + (laddr)), 16, (int)(voff), (int)(soff), 0, 0)
    if ((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem != 0u) __builtin_trap();

    f4_t acc[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f4_t){0.f, 0.f, 0.f, 0.f};

    const bool do_bias = a.dbias != nullptr && jt == 0;
    const int bc = tid % BMC, bh = tid / BMC;
    constexpr int BROWS = 64 / (512 / BMC);
    float bsum = 0.f;

    // Two LDS stages: the tiles of step kb + 1 are in flight while step kb is multiplied; ONE barrier per step (behind the
    // multiplications: by then the next tiles have landed -- vmcnt(0) -- and every wave is done with the current ones).
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
    auto fire = [&](int st) {
#pragma unroll
        for (int j = 0; j < 2; ++j) WG2_BLDS16(yrs, ythr, ysof + (unsigned)j * yrow2, st * STAGE + (32 * j + wid * 4) * 256);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) WG2_BLDS16(xrs, xofs[jj], 0, st * STAGE + YBUF + (jj * 64 + wid * 8) * 128);
    };
    if (kb0 < kb1) {
        prepare(kb0);
        fire(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    int st = 0;
    for (int kb = kb0; kb < kb1; ++kb) {
        if (kb + 1 < kb1) {
            prepare(kb + 1);
            fire(st ^ 1);
        }
        const unsigned sb = (unsigned)(st * STAGE);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf8_t af[MT], bfr[4];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                // channel chunk (wm * BMC/16 + 2 mt + (tp >> 1)) ^ swz(row): the tile index is an XOR of address bits 5-6
                unsigned a0 = ybase[0], a1 = ybase[1];
                if (mt) { WG2_XOR(a0, ybase[0], mt << 5); WG2_XOR(a1, ybase[1], mt << 5); }
                const bf4_t lo = WG2_TR(a0 + sb + kk * 8192);
                const bf4_t hi = WG2_TR(a1 + sb + kk * 8192);
                af[mt] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                unsigned o0 = xbase[kk][0], o1 = xbase[kk][1];
                if (nt) { WG2_XOR(o0, xbase[kk][0], nt << 5); WG2_XOR(o1, xbase[kk][1], nt << 5); }
                const bf4_t lo = WG2_TR(o0 + sb);
                const bf4_t hi = WG2_TR(o1 + sb);
                bfr[nt] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
            if (xrelu) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) bfr[nt] = relu_frag(bfr[nt]);
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
                bsum += bf2f(*(const bf16_t*)(smem + sb + tr_off2(row, bc >> 3) + (bc & 7) * 2));
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        st ^= 1;
    }
    if (do_bias) {
        float* red = (float*)smem;
        red[tid] = bsum;
        __syncthreads();
        if (tid < BMC && co0 + tid < g.Cout) {
            float t = 0.f;
#pragma unroll
            for (int h = 0; h < 512 / BMC; ++h) t += red[tid + h * BMC];
            if (splits == 1 && g.nphase == 1) {
                if (a.overwrite) a.dbias[co0 + tid] = t;
                else a.dbias[co0 + tid] += t;
            } else {
                atomicAdd(a.dbias + co0 + tid, t);
            }
        }
    }
    const int wt = (int)((pg.wt4[ph][q] >> (8 * wn)) & 0xffu);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const size_t cbase = (size_t)wt * g.Cin + ci0 + nt * 16 + fi;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + wm * (BMC / 2) + mt * 16 + fg * 4 + r;
                if (co < g.Cout) {
                    float* pw = a.dw + (size_t)co * g.wtaps * g.Cin + cbase;
                    if (splits == 1) {
                        if (a.overwrite) *pw = acc[mt][nt][r];
                        else *pw += acc[mt][nt][r];
                    } else {
                        atomicAdd(pw, acc[mt][nt][r]);
                    }
                }
            }
        }
    }
}

// ---- host side ----------------------------------------------------------------------------------------------
bool wgrad2_ok(const GG& g) {
    if (!pai_tunable("wgrad2", 0)) return false;    // off by default: measured slower than gg_wgrad_patch_k, see the header
    PatchGeo pg;
    // 32-bit byte offsets into buffer descriptors: every tensor below 2 GB
    if ((int64_t)g.N * g.H * g.W * (g.C1 > g.C2 ? g.C1 : g.C2) * 2 >= (1ll << 31) || (int64_t)g.N * g.OH * g.OW * g.Cout * 2 >= (1ll << 31))
        return false;
    return g.lsw >= 0 && g.lw >= 4 && g.lh >= 2 && (g.C1 % 64) == 0 && (g.C2 % 64) == 0 && (g.Cout % 128) == 0 &&
           g.Cin >= 64 && patch_geo(g, 4, &pg);
}

int launch_wgrad2(const GG& g, const WgradArgs& a, hipStream_t s) {
    PatchGeo pg;
    PAI_CHECK(wgrad2_ok(g) && patch_geo(g, 4, &pg), "launch_wgrad2: problem not eligible");
    const int cotiles = g.Cout / 128;
    const int jtiles = (g.Cin / 64) * pg.groups;
    const int tiles = cotiles * jtiles * g.nphase;
    const int kblocks = g.M / 64;
    // pixel splits: enough workgroups for two per CU, but every split adds one fp32 atomic pass over dW
    int splits = cdiv(pai_tunable("wgrad2_target", 512), tiles);
    const int max_splits = cdiv(g.M, pai_tunable("wgrad2_minrows", 512));
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    const int per = cdiv(kblocks, splits);
    const int psplits = cdiv(kblocks, per);
    const size_t lds = 2 * (64 * 256 + 128 * 128);      // two stages of dY tile + X patch
    PAI_LAUNCH(gg_wgrad_patch2_k<128>, dim3(tiles * psplits), dim3(512), lds, s, g, a, pg, cotiles, jtiles, psplits, per);
    PAI_LAUNCH_CHECK();
    return 0;
}

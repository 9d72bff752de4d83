// bf16 MFMA gather-GEMM kernels for gfx950 (v_mfma_f32_16x16x32_bf16, fp32 accumulate).
//
// Forward / input-gradient:  out[m][co] = sum_{t,ci} A(m,t,ci) * Wp[co][wt][ci]
//   128 x BN x 64 tile, 4 waves (2 x 2), register-staged double-buffered LDS, both operands
//   K-contiguous so every fragment is one ds_read_b128 from an XOR-swizzled 128-B-row image.
// Weight gradient:           dW[co][wt][ci] += sum_m dY[m][co] * A(m,t,ci)
//   both operands are pixel-major in HBM (NHWC), i.e. K-strided: they are staged row-major
//   into LDS and the MFMA fragments are fetched with ds_read_b64_tr_b16 (hardware transpose).
//
// Serves the dense layers of the reference's hot path: EncoderBlock / DecoderBlock convs
// (models/pix2pix.py:58-111), DiscriminatorBlock 1-3 (models/wrapper.py:229-232) and their
// aten::convolution_backward calls.
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf4_t;
typedef __attribute__((ext_vector_type(4))) float f4_t;
typedef __attribute__((ext_vector_type(2))) short s2_t;

constexpr int MBM = 128;  // output rows per block
constexpr int MBK = 64;   // K per iteration (one tap, 64 channels)

int fwd_mfma_mtiles(const GG& g) { return cdiv(g.M, MBM); }

bool fwd_mfma_ok(int dtype, const GG& g, const FwdArgs& a) {
    if (dtype != PAI_BF16) return false;
    if (g.C1 % 64 || g.C2 % 64) return false;
    if (g.Cout % 64) return false;
    if (g.D2 > 0 && (g.D1 % 64)) return false;
    if (a.yf32) return false;
    if (a.skip_d1) return false;
    if ((a.y1 || a.y2) && a.yact) return false;  // one storage-dtype output per launch
    return true;
}

__device__ __forceinline__ uint4 relu_bf16x8(uint4 v) {
    // ReLU on packed bf16: as signed 16-bit integers every negative float is negative
    s2_t z = {0, 0};
    unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        s2_t x = __builtin_bit_cast(s2_t, u[i]);
        x = __builtin_elementwise_max(x, z);
        u[i] = __builtin_bit_cast(unsigned, x);
    }
    return make_uint4(u[0], u[1], u[2], u[3]);
}

__device__ __forceinline__ void decode_row(const GG& g, int m, int& n, int& gy, int& gx) {
    gx = m % g.OWg;
    int r = m / g.OWg;
    gy = r % g.OHg;
    n = r / g.OHg;
}

template <int BN>
__global__ __launch_bounds__(256) void gg_fwd_mfma_k(GG g, FwdArgs a, int mtiles, int ntiles) {
    constexpr int NT = BN / 32;          // 16-col MFMA tiles per wave along N
    constexpr int BJ = BN / 32;          // staging passes for the B tile
    constexpr int A_BYTES = MBM * 128;   // one A buffer
    constexpr int B_BYTES = BN * 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* As = smem;                 // 2 buffers
    unsigned char* Bs = smem + 2 * A_BYTES;   // 2 buffers

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    int bid = blockIdx.x;
    const int bn = bid % ntiles;
    bid /= ntiles;
    const int bm = bid % mtiles;
    const int ph = bid / mtiles;
    const int m0 = bm * MBM, n0 = bn * BN;

    const bf16_t* x1 = (const bf16_t*)a.x1;
    const bf16_t* x2 = (const bf16_t*)a.x2;
    const bf16_t* w = (const bf16_t*)a.w;

    // ---- staging map: 16-B chunk sc of row sr + 32*j ---------------------------
    const int sc = tid & 7, sr = tid >> 3;
    const int sswz = (sr >> 1) & 7;
    const unsigned st_off = (unsigned)(sr * 128 + ((sc ^ sswz) << 4));
    int rnH[4], ry[4], rx[4];
    bool rv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int m = m0 + sr + 32 * j;
        rv[j] = m < g.M;
        int n, gy, gx;
        decode_row(g, rv[j] ? m : 0, n, gy, gx);
        rnH[j] = n * g.H;
        ry[j] = gy * g.S;
        rx[j] = gx * g.S;
    }
    size_t wrow[BJ];
#pragma unroll
    for (int j = 0; j < BJ; ++j) wrow[j] = (size_t)(n0 + sr + 32 * j) * g.wtaps * g.Cin + sc * 8;

    const int cchunks = g.Cin / MBK;
    const int niter = g.ntaps * cchunks;

    uint4 pa[4], pb[BJ];
    auto gload = [&](int it) {
        const int t = it / cchunks;
        const int c0 = (it - t * cchunks) * MBK;
        const int ddy = g.dy[ph][t], ddx = g.dx[ph][t];
        const bf16_t* src;
        int cs, cc, relu;
        if (c0 < g.C1) { src = x1; cs = g.C1; cc = c0; relu = g.relu1; }
        else { src = x2; cs = g.C2; cc = c0 - g.C1; relu = g.relu2; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int iy = ry[j] + ddy, ix = rx[j] + ddx;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (rv[j] && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W) {
                v = *(const uint4*)(src + ((size_t)(rnH[j] + iy) * g.W + ix) * cs + cc + sc * 8);
                if (relu) v = relu_bf16x8(v);
            }
            pa[j] = v;
        }
        const size_t woff = (size_t)g.wt[ph][t] * g.Cin + c0;
#pragma unroll
        for (int j = 0; j < BJ; ++j) pb[j] = *(const uint4*)(w + wrow[j] + woff);
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) *(uint4*)(As + buf * A_BYTES + st_off + j * 32 * 128) = pa[j];
#pragma unroll
        for (int j = 0; j < BJ; ++j) *(uint4*)(Bs + buf * B_BYTES + st_off + j * 32 * 128) = pb[j];
    };

    // ---- fragment read addresses -----------------------------------------------
    const int fr = lane & 15, fq = lane >> 4;
    const int fswz = fr >> 1;  // (row>>1)&7 with row = 16*k + fr
    const unsigned a_base = (unsigned)((wm * 64 + fr) * 128);
    const unsigned b_base = (unsigned)((wn * (BN / 2) + fr) * 128);

    f4_t acc[4][NT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f4_t){0.f, 0.f, 0.f, 0.f};

    gload(0);
    lstore(0);
    __syncthreads();
    for (int it = 0; it < niter; ++it) {
        const int buf = it & 1;
        if (it + 1 < niter) gload(it + 1);
        const unsigned char* Ab = As + buf * A_BYTES;
        const unsigned char* Bb = Bs + buf * B_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const unsigned coff = (unsigned)(((ks * 4 + fq) ^ fswz) << 4);
            bf8_t af[4], bfr[NT];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) af[mt] = *(const bf8_t*)(Ab + a_base + mt * 16 * 128 + coff);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bfr[nt] = *(const bf8_t*)(Bb + b_base + nt * 16 * 128 + coff);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt], bfr[nt], acc[mt][nt], 0, 0, 0);
        }
        if (it + 1 < niter) lstore(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: bias, BN partial statistics, activation, LDS-staged row stores --
    constexpr int CROW = BN * 2 + 16;  // padded bytes per staged output row
    unsigned char* Cs = smem;          // reuse (all LDS reads of the main loop are done)
    float* sstat = (float*)(smem + MBM * CROW);  // [2 wm][2][BN]
    float csum[NT], csq[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int col = wn * (BN / 2) + nt * 16 + fr;
        const float b = a.bias ? a.bias[n0 + col] : 0.f;
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wm * 64 + mt * 16 + fq * 4 + r;
                float v = acc[mt][nt][r] + b;
                if (m0 + row < g.M) { s += v; q += v * v; }
                if (a.yact) v = act_apply(v, a.eact);
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
        dst[0] = sstat[0 * BN + tid] + sstat[2 * BN + tid];
        dst[g.Cout] = sstat[1 * BN + tid] + sstat[3 * BN + tid];
    }
    // destination tensor for this column tile
    bf16_t* dst;
    int dstride, dcol;
    if (a.yact) { dst = (bf16_t*)a.yact; dstride = g.Cout; dcol = n0; }
    else if (n0 < g.D1) { dst = (bf16_t*)a.y1; dstride = g.D1; dcol = n0; }
    else { dst = (bf16_t*)a.y2; dstride = g.D2; dcol = n0 - g.D1; }
    constexpr int CPR = BN / 8;          // 16-B chunks per row
    constexpr int RPP = 256 / CPR;       // rows per pass
    const int oc = tid % CPR, orow0 = tid / CPR;
#pragma unroll
    for (int p = 0; p < MBM / RPP; ++p) {
        const int row = orow0 + p * RPP;
        const int m = m0 + row;
        if (m < g.M) {
            int n, gy, gx;
            decode_row(g, m, n, gy, gx);
            const size_t pix = (size_t)(n * g.OH + gy * g.OS + g.poy[ph]) * g.OW + gx * g.OS + g.pox[ph];
            *(uint4*)(dst + pix * dstride + dcol + oc * 8) = *(const uint4*)(Cs + row * CROW + oc * 16);
        }
    }
}

int launch_fwd_mfma(const GG& g, const FwdArgs& a, hipStream_t s) {
    const int mtiles = fwd_mfma_mtiles(g);
    bool bn128 = (g.Cout % 128) == 0 && (g.D2 == 0 || (g.D1 % 128) == 0);
    if (bn128) {
        const int ntiles = g.Cout / 128;
        const size_t lds = 2 * (MBM * 128 + 128 * 128);
        hipLaunchKernelGGL(gg_fwd_mfma_k<128>, dim3(mtiles * ntiles * g.nphase), dim3(256), lds, s, g, a,
                           mtiles, ntiles);
    } else {
        const int ntiles = g.Cout / 64;
        const size_t lds = 2 * (MBM * 128 + 64 * 128);
        hipLaunchKernelGGL(gg_fwd_mfma_k<64>, dim3(mtiles * ntiles * g.nphase), dim3(256), lds, s, g, a,
                           mtiles, ntiles);
    }
    PAI_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------
// Weight gradient
// ------------------------------------------------------------------------------------
bool wgrad_mfma_ok(int dtype, const GG& g) {
    if (dtype != PAI_BF16) return false;
    if (g.C1 % 64 || g.C2 % 64) return false;
    if (g.Cout % 64) return false;
    if ((g.ntaps * g.Cin) % 128) return false;
    return true;
}

// byte offset of 16-B chunk `ch` (0..15) of row `row` in a 256-B-row LDS image that is
// conflict-free for both ds_write_b128 row stores and ds_read_b64_tr_b16 transposed reads
__device__ __forceinline__ unsigned tr_off(int row, int ch) {
    return (unsigned)(256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))));
}

template <int BMC>  // output-channel tile: 128 or 64
__global__ __launch_bounds__(256) void gg_wgrad_mfma_k(GG g, WgradArgs a, int cotiles, int jtiles,
                                                       int splits, int rows_per_split) {
    constexpr int MT = BMC / 32;           // 16-row MFMA tiles per wave along co
    constexpr int BUF = 64 * 256;          // one staged operand tile: 64 pixel rows x 256 B
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Ys = smem;              // 2 buffers
    unsigned char* Xs = smem + 2 * BUF;    // 2 buffers

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    int bid = blockIdx.x;
    const int jt = bid % jtiles; bid /= jtiles;
    const int cot = bid % cotiles; bid /= cotiles;
    const int split = bid % splits;
    const int ph = bid / splits;
    const int co0 = cot * BMC, j0 = jt * 128;

    const bf16_t* x1 = (const bf16_t*)a.x1;
    const bf16_t* x2 = (const bf16_t*)a.x2;
    const bf16_t* dy = (const bf16_t*)a.dy;

    // staging: chunk sc (0..15) of pixel row sr + 16*j
    const int sc = tid & 15, sr = tid >> 4;
    // this thread's gathered-column chunk -> (tap, source, channel)
    const int jc = j0 + sc * 8;
    const int xt = jc / g.Cin;
    const int xci = jc - xt * g.Cin;
    const int ddy = g.dy[ph][xt], ddx = g.dx[ph][xt];
    const bf16_t* xsrc;
    int xcs, xcc, xrelu;
    if (xci < g.C1) { xsrc = x1; xcs = g.C1; xcc = xci; xrelu = g.relu1; }
    else { xsrc = x2; xcs = g.C2; xcc = xci - g.C1; xrelu = g.relu2; }
    const bool yv = (co0 + sc * 8) < g.Cout;  // BMC = 64 uses only chunks 0..7; Cout tail

    const int mbeg = split * rows_per_split;
    const int mend = min(g.M, mbeg + rows_per_split);
    const int niter = (mend - mbeg + 63) / 64;

    uint4 py[4], px[4];
    auto gload = [&](int it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = mbeg + it * 64 + sr + 16 * j;
            uint4 vy = make_uint4(0, 0, 0, 0), vx = make_uint4(0, 0, 0, 0);
            if (m < mend) {
                int n, gy, gx;
                decode_row(g, m, n, gy, gx);
                if (yv && sc < BMC / 8) {
                    const size_t opix = (size_t)(n * g.OH + gy * g.OS + g.poy[ph]) * g.OW + gx * g.OS + g.pox[ph];
                    vy = *(const uint4*)(dy + opix * g.Cout + co0 + sc * 8);
                }
                const int iy = gy * g.S + ddy, ix = gx * g.S + ddx;
                if (iy >= 0 && iy < g.H && ix >= 0 && ix < g.W) {
                    vx = *(const uint4*)(xsrc + ((size_t)(n * g.H + iy) * g.W + ix) * xcs + xcc);
                    if (xrelu) vx = relu_bf16x8(vx);
                }
            }
            py[j] = vy;
            px[j] = vx;
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned o = tr_off(sr + 16 * j, sc);
            *(uint4*)(Ys + buf * BUF + o) = py[j];
            *(uint4*)(Xs + buf * BUF + o) = px[j];
        }
    };

    // transposed-read lane geometry (guide T10): lane 4q+p of a 16-lane group addresses
    // row r0+q, columns 4p..4p+3 of a 4 x 16 block and receives column (lane&15), rows r0..r0+3
    const int fi = lane & 15, fg = lane >> 4;
    const int tq = fi >> 2, tp = fi & 3;

    f4_t acc[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f4_t){0.f, 0.f, 0.f, 0.f};

    if (niter > 0) {
        gload(0);
        lstore(0);
    }
    __syncthreads();
    for (int it = 0; it < niter; ++it) {
        const int buf = it & 1;
        if (it + 1 < niter) gload(it + 1);
        const unsigned char* Yb = Ys + buf * BUF;
        const unsigned char* Xb = Xs + buf * BUF;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf8_t af[MT], bfr[4];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int row = ks * 32 + fg * 8 + h * 4 + tq;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int ch = (wm * (BMC / 2) + mt * 16) / 8 + (tp >> 1);
                    bf4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (bf4_t __attribute__((address_space(3)))*)(Yb + tr_off(row, ch) + 8 * (tp & 1)));
#pragma unroll
                    for (int e = 0; e < 4; ++e) af[mt][h * 4 + e] = v[e];
                }
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const int ch = (wn * 64 + nt * 16) / 8 + (tp >> 1);
                    bf4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (bf4_t __attribute__((address_space(3)))*)(Xb + tr_off(row, ch) + 8 * (tp & 1)));
#pragma unroll
                    for (int e = 0; e < 4; ++e) bfr[nt][h * 4 + e] = v[e];
                }
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt], bfr[nt], acc[mt][nt], 0, 0, 0);
        }
        if (it + 1 < niter) lstore(buf ^ 1);
        __syncthreads();
    }

    // ---- accumulate into the fp32 gradient (fwd pack) ------------------------------
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int jcol = j0 + wn * 64 + nt * 16 + fi;
        const int t = jcol / g.Cin;
        const int ci = jcol - t * g.Cin;
        const size_t cbase = (size_t)g.wt[ph][t] * g.Cin + ci;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + wm * (BMC / 2) + mt * 16 + fg * 4 + r;
                if (co < g.Cout) atomicAdd(a.dw + (size_t)co * g.wtaps * g.Cin + cbase, acc[mt][nt][r]);
            }
        }
    }
}

int launch_colsum(int dtype, const void* x, int64_t rows, int C, float* out, hipStream_t s);

int launch_wgrad_mfma(const GG& g, const WgradArgs& a, hipStream_t s) {
    const bool big = (g.Cout % 128) == 0;
    const int cotiles = big ? g.Cout / 128 : g.Cout / 64;
    const int jtiles = g.ntaps * g.Cin / 128;
    const int tiles = cotiles * jtiles * g.nphase;
    int splits = cdiv(1024, tiles);
    const int max_splits = cdiv(g.M, 256);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    int rows = cdiv(cdiv(g.M, splits), 64) * 64;
    splits = cdiv(g.M, rows);
    const size_t lds = 4 * 64 * 256;
    dim3 grid(tiles * splits);
    if (big)
        hipLaunchKernelGGL(gg_wgrad_mfma_k<128>, grid, dim3(256), lds, s, g, a, cotiles, jtiles, splits, rows);
    else
        hipLaunchKernelGGL(gg_wgrad_mfma_k<64>, grid, dim3(256), lds, s, g, a, cotiles, jtiles, splits, rows);
    PAI_LAUNCH_CHECK();
    if (a.dbias) {
        // destination pixels of all phases together tile the whole output: plain column sum
        return launch_colsum(PAI_BF16, a.dy, (int64_t)g.N * g.OH * g.OW, g.Cout, a.dbias, s);
    }
    return 0;
}

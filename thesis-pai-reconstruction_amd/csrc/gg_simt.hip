// Gather-GEMM on the vector ALUs (fp32 FMA): the exact-fp32 parity path for every layer,
// and the production path for the HBM-bound edge layers whose GEMM is degenerate
// (Cin in {1,2}: encoders[0] models/pix2pix.py:141-147, discriminator block 0
// models/wrapper.py:229; Cout = 1: decoders[7] models/pix2pix.py:185-193, final
// PatchGAN conv models/wrapper.py:233).
#include "common.h"

// ------------------------------------------------------------------------------------
// LDS-tiled fp32 gather-GEMM, 64x64x16 tile, 256 threads, 4x4 outputs per thread.
// ------------------------------------------------------------------------------------
constexpr int SBM = 64, SBN = 64, SBK = 16;

int fwd_simt_mtiles(const GG& g) { return cdiv(g.M, SBM); }

template <typename T>
__global__ __launch_bounds__(256) void gg_fwd_simt_k(GG g, FwdArgs a) {
    __shared__ float As[SBK][SBM + 4];
    __shared__ float Bs[SBK][SBN + 4];
    __shared__ float red[16][SBN];

    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int ph = blockIdx.z;
    const int m0 = blockIdx.x * SBM, n0 = blockIdx.y * SBN;
    const int K = g.ntaps * g.Cin;
    const T* x1 = (const T*)a.x1;
    const T* x2 = (const T*)a.x2;
    const T* w = (const T*)a.w;

    // loader rows: ty + 16*j
    int rn[4], ry[4], rx[4];
    bool rv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int m = m0 + ty + 16 * j;
        rv[j] = m < g.M;
        int mm = rv[j] ? m : 0;
        rx[j] = (mm % g.OWg) * g.S;
        ry[j] = ((mm / g.OWg) % g.OHg) * g.S;
        rn[j] = mm / (g.OWg * g.OHg);
    }

    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

    for (int k0 = 0; k0 < K; k0 += SBK) {
        const int k = k0 + tx;
        const bool kv = k < K;
        const int t = kv ? k / g.Cin : 0;
        const int ci = kv ? k - t * g.Cin : 0;
        const int ddy = g.dy[ph][t], ddx = g.dx[ph][t], wtap = g.wt[ph][t];
        const T* src;
        int cs, cc, relu;
        if (ci < g.C1) { src = x1; cs = g.C1; cc = ci; relu = g.relu1; }
        else { src = x2; cs = g.C2; cc = ci - g.C1; relu = g.relu2; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = 0.f;
            if (kv && rv[j]) {
                int iy = ry[j] + ddy, ix = rx[j] + ddx;
                if (iy >= 0 && iy < g.H && ix >= 0 && ix < g.W) {
                    v = Conv<T>::ld(src + ((size_t)(rn[j] * g.H + iy) * g.W + ix) * cs + cc);
                    if (relu) v = fmaxf(v, 0.f);
                }
            }
            As[tx][ty + 16 * j] = v;
            int col = n0 + ty + 16 * j;
            float wv = 0.f;
            if (kv && col < g.Cout) wv = Conv<T>::ld(w + ((size_t)col * g.wtaps + wtap) * g.Cin + ci);
            Bs[tx][ty + 16 * j] = wv;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < SBK; ++kk) {
            float av[4], bv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) av[i] = As[kk][ty * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[j] = Bs[kk][tx * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
        }
        __syncthreads();
    }

    // ---- epilogue --------------------------------------------------------------
    float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + ty * 4 + i;
        const bool mv = m < g.M;
        const int mm = mv ? m : 0;
        const int gx = mm % g.OWg, gy = (mm / g.OWg) % g.OHg, n = mm / (g.OWg * g.OHg);
        const size_t pix = (size_t)(n * g.OH + gy * g.OS + g.poy[ph]) * g.OW + gx * g.OS + g.pox[ph];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = n0 + tx * 4 + j;
            if (!mv || col >= g.Cout) continue;
            float v = acc[i][j] + (a.bias ? a.bias[col] : 0.f);
            s[j] += v;
            q[j] += v * v;
            if (col < g.D1) {
                if (a.y1 && !a.skip_d1) Conv<T>::st((T*)a.y1 + pix * g.D1 + col, v);
            } else if (a.y2) {
                Conv<T>::st((T*)a.y2 + pix * g.D2 + (col - g.D1), v);
            }
            if (a.yact || a.yf32) {
                float av = act_apply(v, a.eact);
                if (a.yact) Conv<T>::st((T*)a.yact + pix * g.Cout + col, av);
                if (a.yf32) a.yf32[pix * g.Cout + col] = av;
            }
        }
    }
    if (a.stats) {
        float* dst = a.stats + ((size_t)(ph * gridDim.x + blockIdx.x) * 2) * g.Cout;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 4; ++j) red[ty][tx * 4 + j] = pass ? q[j] : s[j];
            __syncthreads();
            if (tid < SBN) {
                float t = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) t += red[r][tid];
                if (n0 + tid < g.Cout) dst[(size_t)pass * g.Cout + n0 + tid] = t;
            }
        }
    }
}

int launch_fwd_simt(int dtype, const GG& g, const FwdArgs& a, hipStream_t s) {
    dim3 grid(cdiv(g.M, SBM), cdiv(g.Cout, SBN), g.nphase);
    if (dtype == PAI_F32)
        PAI_LAUNCH(gg_fwd_simt_k<float>, grid, dim3(256), 0, s, g, a);
    else
        PAI_LAUNCH(gg_fwd_simt_k<bf16_t>, grid, dim3(256), 0, s, g, a);
    PAI_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------
// Row-dot: Cout <= 2.  A group of 16 lanes owns one output pixel and strides over the
// (tap, channel) axis in 8-channel chunks, so every load is a full contiguous row
// segment; the partial dot products are combined with wave shuffles.
// ------------------------------------------------------------------------------------
bool fwd_rowdot_ok(const GG& g, const FwdArgs& a) {
    return g.Cout <= 2 && (g.C1 % 8) == 0 && (g.C2 % 8) == 0 && !a.stats;
}

template <typename T> struct Vec8;
template <> struct Vec8<float> {
    static __device__ __forceinline__ void ld(const float* p, float* o) {
        float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
        o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w;
        o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
    }
};
template <> struct Vec8<bf16_t> {
    static __device__ __forceinline__ void ld(const bf16_t* p, float* o) {
        uint4 v = *(const uint4*)p;
        unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o[2 * i] = __uint_as_float(u[i] << 16);
            o[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u);
        }
    }
};

constexpr int RD_ROWS_PER_GROUP = 4;

template <typename T, int NCO>
__global__ __launch_bounds__(256) void gg_fwd_rowdot_k(GG g, FwdArgs a) {
    const int tid = threadIdx.x;
    const int lane16 = tid & 15, grp = tid >> 4;
    const int ph = blockIdx.y;
    const T* x1 = (const T*)a.x1;
    const T* x2 = (const T*)a.x2;
    const T* w = (const T*)a.w;
    const int mbase = (blockIdx.x * 16 + grp) * RD_ROWS_PER_GROUP;

    for (int r = 0; r < RD_ROWS_PER_GROUP; ++r) {
        const int m = mbase + r;
        if (m >= g.M) break;  // uniform over the 16-lane group; shuffles below stay inside it
        const int gx = m % g.OWg, gy = (m / g.OWg) % g.OHg, n = m / (g.OWg * g.OHg);
        float acc[NCO];
#pragma unroll
        for (int c = 0; c < NCO; ++c) acc[c] = 0.f;
        for (int t = 0; t < g.ntaps; ++t) {
            const int iy = gy * g.S + g.dy[ph][t], ix = gx * g.S + g.dx[ph][t];
            if (iy < 0 || iy >= g.H || ix < 0 || ix >= g.W) continue;
            const size_t pix = (size_t)(n * g.H + iy) * g.W + ix;
            const int wtap = g.wt[ph][t];
            for (int c0 = lane16 * 8; c0 < g.Cin; c0 += 128) {
                float xv[8];
                if (c0 < g.C1) {
                    Vec8<T>::ld(x1 + pix * g.C1 + c0, xv);
                    if (g.relu1) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) xv[i] = fmaxf(xv[i], 0.f);
                    }
                } else {
                    Vec8<T>::ld(x2 + pix * g.C2 + (c0 - g.C1), xv);
                    if (g.relu2) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) xv[i] = fmaxf(xv[i], 0.f);
                    }
                }
#pragma unroll
                for (int c = 0; c < NCO; ++c) {
                    float wv[8];
                    Vec8<T>::ld(w + ((size_t)c * g.wtaps + wtap) * g.Cin + c0, wv);
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[c] = fmaf(xv[i], wv[i], acc[c]);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < NCO; ++c) {
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) acc[c] += __shfl_xor(acc[c], o, 64);
        }
        if (lane16 == 0) {
            const size_t pix = (size_t)(n * g.OH + gy * g.OS + g.poy[ph]) * g.OW + gx * g.OS + g.pox[ph];
#pragma unroll
            for (int c = 0; c < NCO; ++c) {
                float v = acc[c] + (a.bias ? a.bias[c] : 0.f);
                if (c < g.D1) {
                    if (a.y1 && !a.skip_d1) Conv<T>::st((T*)a.y1 + pix * g.D1 + c, v);
                } else if (a.y2) {
                    Conv<T>::st((T*)a.y2 + pix * g.D2 + (c - g.D1), v);
                }
                if (a.yact || a.yf32) {
                    float av = act_apply(v, a.eact);
                    if (a.yact) Conv<T>::st((T*)a.yact + pix * g.Cout + c, av);
                    if (a.yf32) a.yf32[pix * g.Cout + c] = av;
                }
            }
        }
    }
}

int launch_fwd_rowdot(int dtype, const GG& g, const FwdArgs& a, hipStream_t s) {
    dim3 grid(cdiv(g.M, 16 * RD_ROWS_PER_GROUP), g.nphase);
#define RD_LAUNCH(T, NCO) PAI_LAUNCH((gg_fwd_rowdot_k<T, NCO>), grid, dim3(256), 0, s, g, a)
    if (dtype == PAI_F32) {
        if (g.Cout == 1) RD_LAUNCH(float, 1); else RD_LAUNCH(float, 2);
    } else {
        if (g.Cout == 1) RD_LAUNCH(bf16_t, 1); else RD_LAUNCH(bf16_t, 2);
    }
#undef RD_LAUNCH
    PAI_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------
// Weight gradient, vector-ALU path:  dW[co][wt][ci] += sum_m dY[dest(m)][co] * A(m,t,ci)
// 64(co) x 64(tap,ci) tile, reduction over output pixels in chunks of 16, split over
// blockIdx.z with fp32 atomics into the (pre-zeroed) gradient.
// ------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gg_wgrad_simt_k(GG g, WgradArgs a, int splits, int rows_per_split) {
    __shared__ float Ys[SBK][SBM + 4];
    __shared__ float Xs[SBK][SBN + 4];
    __shared__ float red[16][SBM];

    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int ph = blockIdx.z / splits, split = blockIdx.z % splits;
    const int co0 = blockIdx.x * SBM, j0 = blockIdx.y * SBN;
    const int J = g.ntaps * g.Cin;
    const T* x1 = (const T*)a.x1;
    const T* x2 = (const T*)a.x2;
    const T* dy = (const T*)a.dy;

    // this thread's 4 gathered columns (tap, ci)
    int ct[4], cci[4];
    bool cv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int jc = j0 + tx + 16 * j;
        cv[j] = jc < J;
        ct[j] = cv[j] ? jc / g.Cin : 0;
        cci[j] = cv[j] ? jc - ct[j] * g.Cin : 0;
    }
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    float bsum[4] = {0, 0, 0, 0};

    const int mbeg = split * rows_per_split;
    const int mend = min(g.M, mbeg + rows_per_split);
    for (int mk = mbeg; mk < mend; mk += SBK) {
        const int m = mk + ty;
        const bool mv = m < mend;
        const int mm = mv ? m : 0;
        const int gx = mm % g.OWg, gy = (mm / g.OWg) % g.OHg, n = mm / (g.OWg * g.OHg);
        const size_t opix = (size_t)(n * g.OH + gy * g.OS + g.poy[ph]) * g.OW + gx * g.OS + g.pox[ph];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int co = co0 + tx + 16 * j;
            float v = 0.f;
            if (mv && co < g.Cout) v = Conv<T>::ld(dy + opix * g.Cout + co);
            Ys[ty][tx + 16 * j] = v;
            bsum[j] += v;
            float xv = 0.f;
            if (mv && cv[j]) {
                const int t = ct[j];
                const int iy = gy * g.S + g.dy[ph][t], ix = gx * g.S + g.dx[ph][t];
                if (iy >= 0 && iy < g.H && ix >= 0 && ix < g.W) {
                    const size_t pix = (size_t)(n * g.H + iy) * g.W + ix;
                    const int ci = cci[j];
                    if (ci < g.C1) {
                        xv = Conv<T>::ld(x1 + pix * g.C1 + ci);
                        if (g.relu1) xv = fmaxf(xv, 0.f);
                    } else {
                        xv = Conv<T>::ld(x2 + pix * g.C2 + (ci - g.C1));
                        if (g.relu2) xv = fmaxf(xv, 0.f);
                    }
                }
            }
            Xs[ty][tx + 16 * j] = xv;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < SBK; ++kk) {
            float yv[4], xv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) yv[i] = Ys[kk][ty * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) xv[j] = Xs[kk][tx * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(yv[i], xv[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int co = co0 + ty * 4 + i;
        if (co >= g.Cout) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int jc = j0 + tx * 4 + j;
            if (jc >= J) continue;
            const int t = jc / g.Cin, ci = jc - t * g.Cin;
            atomicAdd(a.dw + ((size_t)co * g.wtaps + g.wt[ph][t]) * g.Cin + ci, acc[i][j]);
        }
    }
    if (a.dbias && blockIdx.y == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) red[ty][tx + 16 * j] = bsum[j];
        __syncthreads();
        if (tid < SBM && co0 + tid < g.Cout) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) t += red[r][tid];
            atomicAdd(a.dbias + co0 + tid, t);
        }
    }
}

int launch_wgrad_simt(int dtype, const GG& g, const WgradArgs& a, hipStream_t s) {
    const int tiles = cdiv(g.Cout, SBM) * cdiv(g.ntaps * g.Cin, SBN) * g.nphase;
    int splits = cdiv(2048, tiles);
    int max_splits = cdiv(g.M, 64);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    int rows = cdiv(g.M, splits);
    rows = cdiv(rows, SBK) * SBK;
    splits = cdiv(g.M, rows);
    dim3 grid(cdiv(g.Cout, SBM), cdiv(g.ntaps * g.Cin, SBN), g.nphase * splits);
    if (dtype == PAI_F32)
        PAI_LAUNCH(gg_wgrad_simt_k<float>, grid, dim3(256), 0, s, g, a, splits, rows);
    else
        PAI_LAUNCH(gg_wgrad_simt_k<bf16_t>, grid, dim3(256), 0, s, g, a, splits, rows);
    PAI_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------
// Weight gradient for Cout <= 2 (decoders[7], final PatchGAN conv): a reduction over
// output pixels of dY[m] * A(m, t, 0..Cin).  Thread = (8-channel chunk, row lane); every
// thread keeps ntaps x 8 partial sums in registers; combined through LDS atomics, then
// one global atomic per weight per block.
// ------------------------------------------------------------------------------------
template <typename T, int NTAPS>
__global__ __launch_bounds__(256) void gg_wgrad_rowdot_k(GG g, WgradArgs a, int rows_per_block) {
    extern __shared__ __attribute__((aligned(16))) float sacc[];  // [NCO][NTAPS][Cin] + NCO
    const int tid = threadIdx.x;
    const int ph = blockIdx.y;
    const int chunks = g.Cin / 8;
    const int chunk = tid % chunks, rl = tid / chunks, nrl = 256 / chunks;
    const int c0 = chunk * 8;
    const T* x1 = (const T*)a.x1;
    const T* x2 = (const T*)a.x2;
    const T* dy = (const T*)a.dy;
    const int nco = g.Cout;
    const int total = nco * NTAPS * g.Cin + nco;
    for (int i = tid; i < total; i += 256) sacc[i] = 0.f;
    __syncthreads();

    const int mbeg = blockIdx.x * rows_per_block;
    const int mend = min(g.M, mbeg + rows_per_block);
    for (int co = 0; co < nco; ++co) {
        float acc[NTAPS][8];
#pragma unroll
        for (int t = 0; t < NTAPS; ++t)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[t][i] = 0.f;
        float bs = 0.f;
        for (int m = mbeg + rl; m < mend; m += nrl) {
            const int gx = m % g.OWg, gy = (m / g.OWg) % g.OHg, n = m / (g.OWg * g.OHg);
            const size_t opix = (size_t)(n * g.OH + gy * g.OS + g.poy[ph]) * g.OW + gx * g.OS + g.pox[ph];
            const float dv = Conv<T>::ld(dy + opix * g.Cout + co);
            if (chunk == 0) bs += dv;
#pragma unroll
            for (int t = 0; t < NTAPS; ++t) {
                const int iy = gy * g.S + g.dy[ph][t], ix = gx * g.S + g.dx[ph][t];
                if (iy < 0 || iy >= g.H || ix < 0 || ix >= g.W) continue;
                const size_t pix = (size_t)(n * g.H + iy) * g.W + ix;
                float xv[8];
                if (c0 < g.C1) {
                    Vec8<T>::ld(x1 + pix * g.C1 + c0, xv);
                    if (g.relu1) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) xv[i] = fmaxf(xv[i], 0.f);
                    }
                } else {
                    Vec8<T>::ld(x2 + pix * g.C2 + (c0 - g.C1), xv);
                    if (g.relu2) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) xv[i] = fmaxf(xv[i], 0.f);
                    }
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[t][i] = fmaf(dv, xv[i], acc[t][i]);
            }
        }
#pragma unroll
        for (int t = 0; t < NTAPS; ++t)
#pragma unroll
            for (int i = 0; i < 8; ++i) atomicAdd(&sacc[(co * NTAPS + t) * g.Cin + c0 + i], acc[t][i]);
        if (chunk == 0) atomicAdd(&sacc[nco * NTAPS * g.Cin + co], bs);
    }
    __syncthreads();
    for (int i = tid; i < nco * NTAPS * g.Cin; i += 256) {
        const int ci = i % g.Cin, t = (i / g.Cin) % NTAPS, co = i / (g.Cin * NTAPS);
        atomicAdd(a.dw + ((size_t)co * g.wtaps + g.wt[ph][t]) * g.Cin + ci, sacc[i]);
    }
    if (a.dbias && tid < nco) atomicAdd(a.dbias + tid, sacc[nco * NTAPS * g.Cin + tid]);
}

int launch_wgrad_rowdot(int dtype, const GG& g, const WgradArgs& a, hipStream_t s) {
    PAI_CHECK(g.ntaps == 4 || g.ntaps == 9 || g.ntaps == 16, "wgrad_rowdot: ntaps %d", g.ntaps);
    int blocks = cdiv(g.M, 256);
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    int rows = cdiv(g.M, blocks);
    blocks = cdiv(g.M, rows);
    dim3 grid(blocks, g.nphase);
    size_t lds = ((size_t)g.Cout * g.ntaps * g.Cin + g.Cout) * sizeof(float);
    PAI_CHECK(lds <= 64 * 1024, "wgrad_rowdot: LDS %zu too large", lds);
#define WR_LAUNCH(T, NT) PAI_LAUNCH((gg_wgrad_rowdot_k<T, NT>), grid, dim3(256), lds, s, g, a, rows)
    if (dtype == PAI_F32) {
        if (g.ntaps == 4) WR_LAUNCH(float, 4); else if (g.ntaps == 9) WR_LAUNCH(float, 9); else WR_LAUNCH(float, 16);
    } else {
        if (g.ntaps == 4) WR_LAUNCH(bf16_t, 4); else if (g.ntaps == 9) WR_LAUNCH(bf16_t, 9); else WR_LAUNCH(bf16_t, 16);
    }
#undef WR_LAUNCH
    PAI_LAUNCH_CHECK();
    return 0;
}

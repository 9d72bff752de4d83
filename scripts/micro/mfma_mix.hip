// What does the instruction mix of gg_fwd_bd_k (gg_bd.hip) cost the matrix pipe, memory latency aside?  (MI355X, gfx950)
// Per wave and "tap": 32 MFMAs 16x16x32 (or 16 MFMAs 32x32x16) on 128 accumulation registers, 8 ds_read_b128 and
// 4 buffer_load_dwordx4 (4 KB, L2-resident), two waves per SIMD, in several arrangements:
//   0  MFMAs only
//   1  as gg_fwd_bd_k: four passes of 8 MFMAs, a buffer load behind each, the 8 LDS reads one per MFMA in the second pass
//   2  all 32 MFMAs, then the 8 LDS reads and 4 buffer loads back to back
//   3  four times: 8 MFMAs, then 2 LDS reads + 1 buffer load back to back
//   4  one LDS read or buffer load behind every 2nd / 3rd MFMA (spread evenly)
// build: hipcc -O2 --offload-arch=gfx950 scripts/micro/mfma_mix.hip -o scripts/micro/mfma_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) float f4_t;
typedef __attribute__((ext_vector_type(16))) float f16_t;
typedef __attribute__((ext_vector_type(4))) unsigned u4_t;

#define MFMA16(acc, a, b)                                                                                  \
    do {                                                                                                   \
        if (ACCV) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));     \
        else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));          \
    } while (0)
#define MFMA32(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))
#define DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
#define BLD(dst, vo, rs, so) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(vo), "s"(rs), "s"(so))

// ACCV: accumulators in the vector registers ("+v") instead of the accumulation registers ("+a")
template <int SHAPE, int PAT, int ACCV = 0>
__global__ __launch_bounds__(256, 2) void k(const uint4* src, float* out, int iters) {
    __shared__ uint4 lds[2048];
    const int tid = threadIdx.x;
    for (int i = tid; i < 2048; i += 256) lds[i] = src[i & 1023];
    __syncthreads();
    const unsigned long long base = (unsigned long long)(size_t)src;
    const u4_t rs = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)base),
                     (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(base >> 32)) & 0xffffu, 16384u, 0x00020000u};
    const unsigned vo = (unsigned)(tid & 63) * 16u, la = (unsigned)(tid & 63) * 16u;
    u4_t w[4], p[8];
    for (int i = 0; i < 4; ++i) w[i] = __builtin_bit_cast(u4_t, src[(tid + 64 * i) & 1023]);
    for (int i = 0; i < 8; ++i) p[i] = __builtin_bit_cast(u4_t, src[(tid * 3 + 64 * i + 7) & 1023]);
    f4_t acc[8][4];
    f16_t acc32[4][2];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (f4_t){0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc32[i][j][r] = 0;
    for (int it = 0; it < iters; ++it) {
        const unsigned so = (unsigned)(it & 3) * 4096u;
        int n = 0;   // running MFMA index of the tap (16x16x32-equivalents)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
                if (SHAPE == 16) MFMA16(acc[mt][nt], w[nt], p[mt]);
                else if ((mt & 1) == 0) MFMA32(acc32[mt >> 1][nt & 1], w[nt], p[mt]);   // one 32x32x16 per two 16x16x32
                ++n;
                if (PAT == 1 && nt == 1) DSR(p[mt], la, 0);
                if (PAT == 4) {
                    if (n % 3 == 0 && n <= 24) DSR(p[(n / 3 - 1) & 7], la, 1024);
                    if (n == 26 || n == 28 || n == 30 || n == 32) BLD(w[(n - 26) / 2], vo, rs, so);
                }
            }
            if (PAT == 1) BLD(w[nt], vo, rs, so);
            if (PAT == 3) { DSR(p[2 * nt], la, 0); DSR(p[2 * nt + 1], la, 2048); BLD(w[nt], vo, rs, so); }
        }
        if (PAT == 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) DSR(p[i], la, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) BLD(w[i], vo, rs, so);
        }
        if (PAT != 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float t = 0;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][3];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) t += acc32[i][j][0] + acc32[i][j][15];
    out[blockIdx.x * 256 + tid] = t + __uint_as_float(w[0][0] ^ p[0][0]);
}

static int g_iters = 2000;   // argv[1]: iterations per launch (a long launch to watch power / clock beside it)
template <int SHAPE, int PAT, int ACCV = 0>
static void run(const uint4* src, float* out) {
    const int iters = g_iters, grid = 256 * 2;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<SHAPE, PAT, ACCV>), dim3(grid), dim3(256), 0, 0, src, out, 100);
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<SHAPE, PAT, ACCV>), dim3(grid), dim3(256), 0, 0, src, out, iters);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flop = (double)grid * 4 * iters * 32.0 * 16 * 16 * 32 * 2;
    printf("mfma %2dx%2d  2 waves/SIMD  %s accumulators  arrangement %d: %7.1f us  %6.0f TFLOP/s\n", SHAPE, SHAPE, ACCV ? "VGPR" : "AGPR", PAT,
           best * 1e3, flop / (best * 1e-3) * 1e-12);
}

int main(int argc, char** argv) {
    if (argc > 1) g_iters = atoi(argv[1]);
    uint4* src; float* out;
    (void)hipMalloc(&src, 1024 * 16); (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    unsigned short h[8192];
    srand(1);
    for (int i = 0; i < 8192; ++i) h[i] = (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));   // bf16 around +-1
    (void)hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    if (argc > 2) {   // argv[2]: only the 16x16x32 loops, bare and with the evenly spread mix, both accumulator homes
        run<16, 0, 0>(src, out); run<16, 0, 1>(src, out); run<16, 4, 0>(src, out); run<16, 4, 1>(src, out);
        return 0;
    }
    run<16, 0>(src, out); run<16, 1>(src, out); run<16, 2>(src, out); run<16, 3>(src, out); run<16, 4>(src, out);
    run<32, 0>(src, out); run<32, 1>(src, out); run<32, 2>(src, out); run<32, 3>(src, out); run<32, 4>(src, out);
    return 0;
}

// How many other vector instructions per MFMA can a SIMD issue before the matrix pipe starves?  (MI355X, gfx950)
//   mfma_issue            -> table: MFMA shape x waves per SIMD x {0..4 v_pk_max_i16, 1-2 ds_read_b128} per MFMA -> TFLOP/s
// Each wave keeps a 64 x 64 fp32 accumulator tile (64 VGPRs) like the convolution kernels and issues MFMAs on register
// operands; the fillers are independent of the MFMAs (no data hazard), so what is measured is issue bandwidth and the
// clock the chip holds under that load.
// build: hipcc -O2 --offload-arch=gfx950 scripts/micro/mfma_issue.hip -o scripts/micro/mfma_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf8_t;
typedef __attribute__((ext_vector_type(4))) float f4_t;
typedef __attribute__((ext_vector_type(16))) float f16_t;

template <int SHAPE, int NV, int NL>
__global__ __launch_bounds__(256) void k(const uint4* src, float* out, int iters) {
    __shared__ uint4 lds[1024];
    const int tid = threadIdx.x;
    for (int i = tid; i < 1024; i += 256) lds[i] = src[i];
    __syncthreads();
    bf8_t a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = __builtin_bit_cast(bf8_t, src[(tid + 64 * i) & 1023]);
        b[i] = __builtin_bit_cast(bf8_t, src[(tid * 3 + 64 * i + 7) & 1023]);
    }
    int f0 = tid, f1 = tid * 5, f2 = tid * 7, f3 = tid * 9;
    uint4 l0 = make_uint4(0, 0, 0, 0), l1 = l0;
    const unsigned la = (unsigned)(tid & 63) * 16u;
    if (SHAPE == 16) {
        f4_t acc[4][4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (f4_t){0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
                    if (NV > 0) asm volatile("v_pk_max_i16 %0, %0, 0" : "+v"(f0));
                    if (NV > 1) asm volatile("v_pk_max_i16 %0, %0, 0" : "+v"(f1));
                    if (NV > 2) asm volatile("v_pk_max_i16 %0, %0, 0" : "+v"(f2));
                    if (NV > 3) asm volatile("v_pk_max_i16 %0, %0, 0" : "+v"(f3));
                    if (NL > 0) asm volatile("ds_read_b128 %0, %1" : "=v"(l0) : "v"(la));
                    if (NL > 1) asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(l1) : "v"(la));
                }
            if (NL > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        float t = 0;
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][3];
        out[blockIdx.x * 256 + tid] = t + f0 + f1 + f2 + f3 + l0.x + l1.y;
    } else {
        f16_t acc[2][2];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i + 2 * kk], b[j + 2 * kk], acc[i][j], 0, 0, 0);
                        // same fillers per FLOP as the 16x16x32 arm: 2 NV / 2 NL per (twice as large) MFMA
#pragma unroll
                        for (int r = 0; r < 2; ++r) {
                            if (NV > 0) asm volatile("v_pk_max_i16 %0, %0, 0" : "+v"(f0));
                            if (NV > 1) asm volatile("v_pk_max_i16 %0, %0, 0" : "+v"(f1));
                            if (NV > 2) asm volatile("v_pk_max_i16 %0, %0, 0" : "+v"(f2));
                            if (NV > 3) asm volatile("v_pk_max_i16 %0, %0, 0" : "+v"(f3));
                            if (NL > 0) asm volatile("ds_read_b128 %0, %1" : "=v"(l0) : "v"(la));
                            if (NL > 1) asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(l1) : "v"(la));
                        }
                    }
            if (NL > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        float t = 0;
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) t += acc[i][j][0] + acc[i][j][15];
        out[blockIdx.x * 256 + tid] = t + f0 + f1 + f2 + f3 + l0.x + l1.y;
    }
}

template <int SHAPE, int NV, int NL>
static void run(const uint4* src, float* out, int wgs_per_cu) {
    const int iters = 4000, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<SHAPE, NV, NL>), dim3(grid), dim3(256), 0, 0, src, out, 200);
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<SHAPE, NV, NL>), dim3(grid), dim3(256), 0, 0, src, out, iters);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    // per iteration and wave: 16 MFMAs of 16x16x32 or 8 of 32x32x16 = 262144 FLOP x 2
    const double flop = (double)grid * 4 * iters * 16.0 * 16 * 16 * 32 * 2;
    printf("mfma %2dx%2d  %d waves/SIMD  %d valu + %d ds_read_b128 per 16x16x32-equivalent: %7.1f us  %6.0f TFLOP/s\n", SHAPE, SHAPE,
           wgs_per_cu, NV, NL, best * 1e3, flop / (best * 1e-3) * 1e-12);
}

int main() {
    uint4* src; float* out;
    (void)hipMalloc(&src, 1024 * 16); (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    unsigned short h[8192];
    srand(1);
    for (int i = 0; i < 8192; ++i) h[i] = (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));   // bf16 around +-1
    (void)hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    for (int w = 1; w <= 4; w *= 2) {
        run<16, 0, 0>(src, out, w); run<16, 1, 0>(src, out, w); run<16, 2, 0>(src, out, w); run<16, 3, 0>(src, out, w); run<16, 4, 0>(src, out, w);
        run<16, 0, 1>(src, out, w); run<16, 2, 1>(src, out, w); run<16, 0, 2>(src, out, w);
        run<32, 0, 0>(src, out, w); run<32, 1, 0>(src, out, w); run<32, 2, 0>(src, out, w); run<32, 3, 0>(src, out, w); run<32, 4, 0>(src, out, w);
        run<32, 0, 1>(src, out, w); run<32, 2, 1>(src, out, w); run<32, 0, 2>(src, out, w);
    }
    return 0;
}

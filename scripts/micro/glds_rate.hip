// LDS-DMA fill rate per CU: 256-thread blocks, each iteration = 8 global_load_lds_dwordx4 per thread
// (32 KB per block) from a source window of `span` bytes (32 KB: L1-resident, 2 MB: L2, 1 GB: HBM).
// build: hipcc -O3 --offload-arch=gfx950 -o glds_rate glds_rate.hip ; run: ./glds_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define GLDS16(gptr, lptr)                                                                      \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),    \
                                     (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

template <int MODE>  // 0: wait + barrier every tile (single buffer), 1: two tiles in flight, 2: + 16 ds_read_b128 per wave per tile
__global__ __launch_bounds__(256) void fill_k(const unsigned char* src, size_t span_mask, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
    size_t base = ((size_t)blockIdx.x * 32768 * 7) & span_mask;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        unsigned char* dst = smem + (MODE ? (it & 1) * 32768 : 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const size_t off = (base + (size_t)j * 4096 + wid * 1024 + lane * 16) & span_mask;
            GLDS16(src + off, dst + j * 4096 + wid * 1024);
        }
        base = (base + 32768) & span_mask;
        if (MODE == 0) {
            __syncthreads();
        } else {
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        if (MODE == 2) {
            const unsigned char* rd = smem + ((it + 1) & 1) * 32768;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float4 v = *(const float4*)(rd + ((r * 1024 + lane * 16 + wid * 8192) & 32767));
                acc += v.x + v.w;
            }
        }
        if (MODE == 0) __syncthreads();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 12345.f) sink[0] = acc + smem[tid];
}

int main() {
    const size_t cap = (size_t)1 << 30;
    unsigned char* src;
    float* sink;
    hipMalloc(&src, cap);
    hipMemset(src, 1, cap);
    hipMalloc(&sink, 4);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    hipFuncSetAttribute((const void*)fill_k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)fill_k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    const int iters = 2000;
    for (int mode = 0; mode < 3; ++mode)
        for (size_t span : {(size_t)32768, (size_t)2 << 20, (size_t)64 << 20, cap})
            for (int bpc : {1, 2, 4}) {
                if (mode && bpc > 2) continue;
                const int blocks = cus * bpc;
                hipEvent_t e0, e1;
                hipEventCreate(&e0);
                hipEventCreate(&e1);
                auto launch = [&]() {
                    if (mode == 0) hipLaunchKernelGGL(fill_k<0>, dim3(blocks), dim3(256), 32768, 0, src, span - 1, iters, sink);
                    if (mode == 1) hipLaunchKernelGGL(fill_k<1>, dim3(blocks), dim3(256), 65536, 0, src, span - 1, iters, sink);
                    if (mode == 2) hipLaunchKernelGGL(fill_k<2>, dim3(blocks), dim3(256), 65536, 0, src, span - 1, iters, sink);
                };
                launch();
                hipEventRecord(e0);
                launch();
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                const double bytes = (double)blocks * iters * 32768;
                printf("mode %d span %7zu KB blocks/CU %d: %7.2f TB/s chip, %6.1f GB/s per CU\n", mode, span >> 10, bpc,
                       bytes / ms / 1e9, bytes / ms / 1e6 / cus);
            }
    return 0;
}

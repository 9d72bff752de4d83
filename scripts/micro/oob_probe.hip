#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
// Does an out-of-range lane of `buffer_load_dwordx4 ... offen lds` write zeros to LDS or leave it untouched?
__global__ void k(const unsigned* src, unsigned nbytes, unsigned* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned* s = (unsigned*)smem;
    for (int i = threadIdx.x; i < 256; i += 64) s[i] = 0xdeadbeefu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    unsigned off = threadIdx.x * 16;
    if (threadIdx.x % 3 == 1) off = 0xfffffff0u;         // out of range
    if (threadIdx.x % 3 == 2) off = nbytes;              // first byte beyond the buffer
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)smem, 16, off, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = s[i];
}
int main() {
    unsigned *src, *out, h[256], hs[256];
    for (int i = 0; i < 256; ++i) hs[i] = 0x1000 + i;
    hipMalloc(&src, 1024); hipMalloc(&out, 1024);
    hipMemcpy(src, hs, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, src, 1024u, out);
    hipMemcpy(h, out, 1024, hipMemcpyDeviceToHost);
    for (int l = 0; l < 6; ++l) printf("lane %d: %08x %08x %08x %08x\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    int zeros = 0, stale = 0, ok = 0;
    for (int l = 0; l < 64; ++l) {
        if (l % 3 == 0) ok += h[l * 4] == 0x1000u + l * 4;
        else { zeros += h[l * 4] == 0; stale += h[l * 4] == 0xdeadbeefu; }
    }
    printf("in-range correct %d/22, out-of-range lanes: zero %d, stale %d of 42\n", ok, zeros, stale);
    return 0;
}

// HBM rate of 16-B-per-lane loads / stores whose 128-B pixel segments are CONTIGUOUS against every second segment (the
// access shape of one output phase of a 4-phase (stride-2) input gradient with 64 channels: 128 B per pixel at a 256-B
// pitch, the other phase's workgroup touches the segments in between at another time).   (round 6)
// build: hipcc -O3 --offload-arch=gfx950 scripts/micro/stride_probe.hip -o scripts/micro/stride_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// mode 0: read, 1: write, 2: read + write (copy).  stride2: segment index s -> byte (2 s + half) * 128 (two passes: half 0, 1)
template <int MODE>
__global__ __launch_bounds__(256) void k(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t nseg, int stride2, int half, uint4* sink) {
    uint4 acc = make_uint4(0, 0, 0, 0);
    const size_t lane8 = threadIdx.x & 7, seg0 = (size_t)blockIdx.x * 32 + (threadIdx.x >> 3);
    for (size_t s = seg0; s < nseg; s += (size_t)gridDim.x * 32) {
        const size_t off16 = (stride2 ? (2 * s + half) : s) * 8 + lane8;      // in 16-B units
        if (MODE != 1) { uint4 v = src[off16]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; if (MODE == 2) dst[off16] = v; }
        else dst[off16] = make_uint4((unsigned)s, 1, 2, 3);
    }
    if (acc.x == 0x12345678u) *sink = acc;
}

int main() {
    const size_t bytes = (size_t)1 << 30, n16 = bytes / 16, nsegAll = bytes / 128;
    uint4 *a, *b, *sink;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 8192;
    for (int mode = 0; mode < 3; ++mode)
        for (int stride2 = 0; stride2 < 2; ++stride2) {
            const size_t nseg = stride2 ? nsegAll / 2 : nsegAll / 2;      // the same number of bytes either way (half the buffer)
            float best = 1e30f;
            for (int it = 0; it < 6; ++it) {
                CK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, a, b, nseg, stride2, it & 1, sink);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, a, b, nseg, stride2, it & 1, sink);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, a, b, nseg, stride2, it & 1, sink);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (it > 0 && ms < best) best = ms;
            }
            const double gb = (double)nseg * 128 * (mode == 2 ? 2 : 1) / 1e9;
            printf("%s %-22s %7.1f us  %5.2f TB/s\n", mode == 0 ? "read " : mode == 1 ? "write" : "copy ", stride2 ? "128 B at a 256-B pitch" : "contiguous",
                   best * 1e3, gb / (best * 1e-3) / 1e3);
        }
    (void)n16;
    return 0;
}

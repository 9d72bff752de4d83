// How fast does a wave write a [M][64] bf16 matrix (128-B rows) by store pattern, cold (a 1 GB fill in front of every
// timed launch) and warm?  mode 0: per instruction 64 lanes x 16 B CONTIGUOUS (8 whole rows = 1 KB); mode 1: the thin
// kernels' pattern -- per instruction one 64-B half of each of 16 rows (lanes fq = 0..3 of pixel fr), the other halves
// with the next instruction; mode 2: mode 1 with the two instructions covering rows 0-7 / 8-15 whole (16 B per lane at
// (lane / 8) * 128 + (lane % 8) * 16 -- what a cross-lane transpose would give).
// build: hipcc -O3 --offload-arch=gfx950 scripts/micro/store_pattern.hip -o scripts/micro/store_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void wr(uint4* out, long rows) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const long step = (long)gridDim.x * 64;
    const uint4 v = make_uint4(lane, wid, blockIdx.x, 7);
    for (long r0 = ((long)blockIdx.x * 4 + wid) * 16; r0 < rows; r0 += step) {
        char* base = (char*)out + r0 * 128;
        if (MODE == 0 || MODE == 2) {
            *(uint4*)(base + lane * 16) = v;
            *(uint4*)(base + 1024 + lane * 16) = v;
        } else {
            *(uint4*)(base + fr * 128 + fq * 16) = v;
            *(uint4*)(base + fr * 128 + 64 + fq * 16) = v;
        }
    }
}
__global__ void fill(uint4* p, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) p[i] = make_uint4(1, 2, 3, 4);
}

int main() {
    const long rows = 2097152;      // D block 0 at batch 128: 2 M rows of 128 B = 268 MB
    uint4 *out, *junk;
    CK(hipMalloc(&out, rows * 128));
    CK(hipMalloc(&junk, 1l << 30));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int cold = 0; cold < 2; ++cold)
        for (int mode = 0; mode < 2; ++mode)
            for (int blocks : {1024, 4096, 16384}) {
                float best = 1e9f;
                for (int it = 0; it < 6; ++it) {
                    if (cold) fill<<<4096, 256>>>(junk, (1l << 30) / 16);
                    CK(hipEventRecord(e0));
                    if (mode == 0) wr<0><<<blocks, 256>>>(out, rows); else wr<1><<<blocks, 256>>>(out, rows);
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (it > 0 && ms < best) best = ms;
                }
                printf("%s mode %d blocks %5d: %7.1f us  %5.2f TB/s\n", cold ? "cold" : "warm", mode, blocks, best * 1e3, rows * 128 / (best * 1e-3) / 1e12);
            }
    return 0;
}

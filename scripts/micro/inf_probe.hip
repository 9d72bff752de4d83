// What v > 0 ? v : fma(v, slope, 0) gives for v = -inf, slope = 0 on the chip, beside the selecting form of common.h.
// hipcc -O3 --offload-arch=gfx950 -o scripts/micro/inf_probe scripts/micro/inf_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
__device__ __forceinline__ float act_mul(float v, float slope) { const float sv = fmaf(v, slope, 0.f); return v > 0.f ? v : sv; }
__device__ __forceinline__ float act_sel(float v, float slope) { const float sv = slope == 0.f ? 0.f : fmaf(v, slope, 0.f); return v > 0.f ? v : sv; }
__global__ void k(const float* in, float* a, float* b, float slope) {
    a[threadIdx.x] = act_mul(in[threadIdx.x], slope);
    b[threadIdx.x] = act_sel(in[threadIdx.x], slope);
}
int main() {
    float h[4] = {-INFINITY, INFINITY, -1.f, NAN}, ha[4], hb[4], *d, *a, *b;
    hipMalloc(&d, 16); hipMalloc(&a, 16); hipMalloc(&b, 16);
    hipMemcpy(d, h, 16, hipMemcpyHostToDevice);
    for (float slope : {0.f, 0.2f}) {
        k<<<1, 4>>>(d, a, b, slope);
        hipMemcpy(ha, a, 16, hipMemcpyDeviceToHost); hipMemcpy(hb, b, 16, hipMemcpyDeviceToHost);
        for (int i = 0; i < 4; ++i) printf("slope %.1f v %5g: fma form %5g  select form %5g\n", slope, h[i], ha[i], hb[i]);
    }
    return 0;
}

// Shared device/host helpers for libpai_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/pai_hip.h"
#include "plan.h"

typedef unsigned short bf16_t;  // raw bfloat16 bits

void pai_set_error(const char* fmt, ...);

#define PAI_CHECK(cond, ...)                 \
    do {                                     \
        if (!(cond)) {                       \
            pai_set_error(__VA_ARGS__);      \
            return 1;                        \
        }                                    \
    } while (0)

#define PAI_LAUNCH_CHECK()                                              \
    do {                                                                \
        hipError_t e_ = hipGetLastError();                              \
        if (e_ != hipSuccess) {                                         \
            pai_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, \
                          hipGetErrorString(e_));                       \
            return 2;                                                   \
        }                                                               \
    } while (0)

// ---- bf16 <-> f32 -----------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
// round-to-nearest-even, NaN stays quiet NaN: v_cvt_pk_bf16_f32 (one instruction; the integer formulation
// took ~7 and dominated the store epilogues of the thin layers)
typedef __attribute__((ext_vector_type(2))) __bf16 pai_bf2_t;
typedef __attribute__((ext_vector_type(2))) float pai_f2_t;
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
// two values -> one packed dword (lo = a, hi = b)
// Activations in a store without run-time branches.  act(v) = v > 0 ? v : v * slope with slope = 1 (none), 0.2
// (LeakyReLU), 0 (ReLU): the selector used to be a run-time switch per ELEMENT, which hipcc compiled into four scalar
// branches per element -- the epilogue of a 256 x 128 tile with the fused producer backward was 4200 instructions per
// thread, more issue time than the 512 MFMAs of a K = 1024 layer (profiles/r06_isa_census.txt).  The product is formed as
// fma(v, slope, +0): the same rounding as the multiplication, and -0 (ReLU of a negative value) becomes +0.  ReLU of
// -inf must be 0 (aten), not -inf x 0 = NaN: slope == 0 selects a literal zero (a wave-uniform condition: one v_cndmask).
__device__ __forceinline__ float act_slope(int act) { return act == PAI_ACT_RELU ? 0.f : (act == PAI_ACT_LRELU ? 0.2f : 1.f); }
__device__ __forceinline__ float act_fwd(float v, float slope) {
    const float sv = slope == 0.f ? 0.f : fmaf(v, slope, 0.f);
    return !(v <= 0.f) ? v : sv;         // a NaN stays a NaN under ReLU too (aten's clamp_min), at no instruction more
}
__device__ __forceinline__ unsigned pk2bf(float a, float b) {
    const pai_f2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, pai_bf2_t));
}

template <typename T> struct Conv;
template <> struct Conv<float> {
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Conv<bf16_t> {
    static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
    static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

// ---- 8-wide vector access ----------------------------------------------------------------
template <typename T> struct V8;
template <> struct V8<float> {
    static __device__ __forceinline__ void ld(const float* p, float* o) {
        float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
        o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
    }
    static __device__ __forceinline__ void st(float* p, const float* v) {
        *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
        *(float4*)(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
};
template <> struct V8<bf16_t> {
    static __device__ __forceinline__ void ld(const bf16_t* p, float* o) {
        uint4 v = *(const uint4*)p;
        unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o[2 * i] = __uint_as_float(u[i] << 16);
            o[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ void st(bf16_t* p, const float* v) {
        unsigned u[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) u[i] = pk2bf(v[2 * i], v[2 * i + 1]);
        *(uint4*)p = make_uint4(u[0], u[1], u[2], u[3]);
    }
};

__device__ __forceinline__ float act_apply(float v, int act) {
    switch (act) {
        case PAI_ACT_LRELU: return v > 0.f ? v : 0.2f * v;
        case PAI_ACT_RELU: return v > 0.f ? v : 0.f;
        case PAI_ACT_TANH: return tanhf(v);
        default: return v;
    }
}
// derivative expressed through the sign of the stored (activated or raw) value
__device__ __forceinline__ float act_grad(float a, int act) {
    switch (act) {
        case PAI_ACT_LRELU: return a > 0.f ? 1.f : 0.2f;
        case PAI_ACT_RELU: return a > 0.f ? 1.f : 0.f;
        default: return 1.f;
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// "Once per DEVICE" flag for hipFuncSetAttribute(MaxDynamicSharedMemorySize): the attribute is per device, a process-wide
// static bool leaves the second GPU of a process without it (the launch then fails: > 64 KB of dynamic LDS).
struct PerDeviceOnce {
    bool done[32] = {};
    // true when the caller has to set the attributes for the current device now
    bool first() {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return true;   // unknown device: set every time (cheap)
        if (done[dev]) return false;
        done[dev] = true;
        return true;
    }
};

// ---- gather-GEMM problem ------------------------------------------------------
// out[m, co] = sum_{t < ntaps} sum_{ci < Cin} A(m, t, ci) * Wp[co][wt[ph][t]][ci]
//   m = (n, gy, gx) over the per-phase output grid OHg x OWg
//   A(m,t,ci) = act(src[n][gy*S + dy[ph][t]][gx*S + dx[ph][t]][ci]) (0 outside)
//   src = x1 for ci < C1, x2 otherwise
//   destination pixel = (gy*OS + poy[ph], gx*OS + pox[ph]) of an OH x OW image,
//   destination channels split D1 | D2 over two tensors.
struct GG {
    int N, H, W;
    int C1, C2, Cin;
    int OHg, OWg;
    int Cout;
    int S;
    int nphase, ntaps;
    int OH, OW, OS;
    int D1, D2;
    int wtaps;
    int relu1, relu2;
    int M;  // rows per phase = N*OHg*OWg
    int lw, lh;  // log2(OWg), log2(OHg) when both are powers of two, else -1
    int lsw, lsh, ldw, ldh;  // log2 of source W, H and destination OW, OH (all -1 unless all are powers of two)
    signed char dy[4][16], dx[4][16], wt[4][16];
    signed char poy[4], pox[4];
    int gslice;  // > 0: block-diagonal filter (pai_conv_desc.groups): 16-channel slices are independent
    int solo;    // pai_conv_desc.hints & PAI_HINT_SOLO: nothing else runs beside this launch (launch geometry only)
};

// forward gather of a pai_conv_desc (Conv2d or ConvTranspose2d)
int gg_build_fwd(const pai_conv_desc* d, GG* g);
// input-gradient gather: source = dy of the layer, destination = dx
int gg_build_dgrad(const pai_conv_desc* d, GG* g);

// launchers (implemented in the .hip files)
struct FwdArgs {
    const void *x1, *x2, *w;
    const float* bias;
    void *y1, *y2;  // raw output (storage dtype), split D1|D2
    void* yact;     // activated output (storage dtype)
    float* yf32;    // activated output fp32
    float* stats;   // [nphase*mtiles][2][Cout]
    int eact;
    int skip_d1;    // only write the D2 part (pai_conv_dgrad only_c2)
    // Backward of the layer that produced x1, fused into the store of the D1 output (pai_conv_dgrad_act,
    // pai_conv_dgrad_bn):  dx1 = bact1'(pre) * dgrad + bact2'(pre) * badd,  pre = bz * bscale + bshift
    // (bscale == NULL: pre = bz, the stored activation).  With bpart the BatchNorm-backward partial sums
    // [rows][2][D1] = (sum dx1, sum dx1 * (bz - bmean) * brstd) are written, one row per output tile.
    const void* bz;
    const void* badd;
    const float *bscale, *bshift, *bmean, *brstd;
    int bact1, bact2;
    float* bpart;
    // Prologue (pai_conv_fwd_pro): x1 is read as pact(x1 * pscale[c] + pshift[c]) -- the BatchNorm + activation of the layer
    // that produced it, applied on load (zero padding stays zero).  Kernels that take it: pwx_k, grouped3_k.
    const float *pscale, *pshift;
    int pact;
};
// BatchNorm finalize + apply in one launch for small layers (bn.hip)
bool bn_fuse_small_ok(int rows, int64_t M, int C);
int launch_bn_fin_apply(int dtype, const float* stats, int rows, int C, int64_t count, const float* gamma, const float* beta,
                        float eps, float momentum, int n_updates, float* running_mean, float* running_var, int64_t* nbt,
                        float* mean, float* rstd, float* scale, float* shift, const void* z, int act, void* out,
                        hipStream_t s);
int launch_bn_bwd_fin_apply(int dtype, const float* partials, int rows, int C, float* sums, float* dgamma, float* dbeta,
                            const void* du, const void* z, int64_t M, const float* mean, const float* rstd,
                            const float* gamma, void* dz, hipStream_t s);
// streaming forms of the big bf16 tensor passes (ew_stream.hip): 0 = launched, > 0 = error, -1 = not taken
int ew_stream_bn_apply(int dtype, const void* z, int64_t M, int C, const float* scale, const float* shift, int act, void* out,
                       hipStream_t s);
int ew_stream_bn2_add_act(int dtype, const void* za, const float* sca, const float* sha, const void* zb, const float* scb,
                          const float* shb, int64_t M, int C, int act_a, int act, void* out, hipStream_t s);
int ew_stream_add_act(int dtype, const void* a, const void* b, int64_t numel, int act, void* out, hipStream_t s);
int ew_stream_bn_bwd_reduce(int dtype, const void* g, int act, const void* z, int64_t M, int C, const float* scale,
                            const float* shift, const float* mean, const float* rstd, float* partials, int rows, hipStream_t s);
int ew_stream_bn_bwd_apply(int dtype, const void* g, int act, const void* z, int64_t M, int C, const float* scale,
                           const float* shift, const float* mean, const float* rstd, const float* gamma, const float* sums,
                           void* dz, hipStream_t s);
int ew_stream_bn2_bwd_reduce(int dtype, const void* d, int act_a, const void* za, const void* zb, int64_t M, int C,
                             const float* scale_a, const float* shift_a, const float* mean_a, const float* rstd_a,
                             const float* mean_b, const float* rstd_b, float* part_a, float* part_b, int rows, hipStream_t s);
int ew_stream_bn2_bwd_apply(int dtype, const void* d, int act_a, const void* za, const void* zb, int64_t M, int C,
                            const float* scale_a, const float* shift_a, const float* mean_a, const float* rstd_a,
                            const float* gamma_a, const float* sums_a, const float* mean_b, const float* rstd_b,
                            const float* gamma_b, const float* sums_b, void* dza, void* dzb, hipStream_t s);
int fwd_mfma_ksplit_effective(const GG& g);     // the K split launch_fwd_mfma uses with the registered workspace
int launch_fwd_simt(int dtype, const GG& g, const FwdArgs& a, hipStream_t s);
int launch_fwd_rowdot(int dtype, const GG& g, const FwdArgs& a, hipStream_t s);
int launch_fwd_mfma(const GG& g, const FwdArgs& a, hipStream_t s);
// run-time tunables (pai_set_tunable): kernel-selection switches for A/B timing and for tests that pin a kernel
int pai_tunable(const char* name, int def);
int fwd_mfma_ksplit(const GG& g);
const char* fwd_mfma_kernel_name(const GG& g);
const char* wgrad_mfma_kernel_name(const GG& g);
bool wgrad_pro_ok(int dtype, const GG& g);
int64_t fwd_mfma_workspace_bytes(const GG& g);
// Per-device handle (pai_create / pai_bind): owner of the caller-provided split-K workspace and general scratch.
// Entry points use the ACTIVE handle of the current HIP device (pai_ctx()); there is no process-wide buffer.
struct pai_handle_s {
    int device;
    float* workspace;          // split-K partial sums (fp32 slabs)
    int64_t workspace_bytes;
    float* scratch;            // thin layers: skinny-GEMM output (head of the buffer), weight-gradient partial tiles (tail)
    int64_t scratch_bytes;
    float* wslab;              // weight gradients split over the pixels: one fp32 dW slab per split (gg_wg3.hip)
    int64_t wslab_bytes;
};
// active handle of the calling thread's current HIP device; never null (a device without a handle has an empty one:
// no workspace, no scratch -> un-split / fallback kernels)
const pai_handle_s* pai_ctx();
int fwd_simt_mtiles(const GG& g);
int fwd_mfma_mtiles(const GG& g);
bool fwd_mfma_ok(int dtype, const GG& g, const FwdArgs& a);
// skinny pointwise convolution (64 <-> 32 channels) on the matrix cores, no LDS (gg_mfma.hip)
bool pw_ok(int dtype, const GG& g, const FwdArgs& a);
int pw_rows(const GG& g);
int launch_pw(const GG& g, const FwdArgs& a, hipStream_t s);
// streaming pointwise convolution of the residual blocks' fine levels (gg_pw.hip)
bool pwx_ok(int dtype, const GG& g, const FwdArgs& a);
int pwx_rows(const GG& g);
int launch_pwx(const GG& g, const FwdArgs& a, hipStream_t s);
const char* pwx_kernel_name(const GG& g);
bool fwd_rowdot_ok(const GG& g, const FwdArgs& a);
// 16- / 32-channel 1x1 and 3x3 convolutions on the matrix cores, no LDS (gg_small.hip)
bool small_ok(int dtype, const GG& g, const FwdArgs& a);
int small_rows(const GG& g);
int launch_small(const GG& g, const FwdArgs& a, hipStream_t s);
// grouped 3x3 convolution, patch in LDS, 16-channel slices on the matrix cores (gg_group.hip)
bool grouped3_ok(int dtype, const GG& g, const FwdArgs& a);
int grouped3_rows(const GG& g);
int launch_grouped3(const GG& g, const FwdArgs& a, hipStream_t s);

// thin layers on the matrix cores (gg_thin.hip)
bool thin_fwd_ok(int dtype, const GG& g, const FwdArgs& a);
bool thin_dgrad_ok(int dtype, const GG& g, const FwdArgs& a);
bool thin_dgrad_shape_ok(int dtype, const GG& g);
int64_t thin_dgrad_scratch_bytes(const GG& g, const FwdArgs& a);
int launch_thin_fwd(const GG& g, const FwdArgs& a, hipStream_t s);
// input gradient of a many-channel -> 1 convolution (the PatchGAN head), activation derivative in the store (gg_thin.hip)
bool head_dgrad_ok(int dtype, const GG& g, const FwdArgs& a);
int launch_head_dgrad(const GG& g, const FwdArgs& a, hipStream_t s);
int thin_fwd_bwd_rows(int dtype, const GG& g, const FwdArgs& a, int act1, const void* add, const float* scale);
int launch_thin_dgrad(const GG& g, const FwdArgs& a, hipStream_t s);

struct WgradArgs {
    const void *x1, *x2, *dy;
    float* dw;
    float* dbias;
    int overwrite;   // dw = instead of +=; honoured by gg_wgrad_mfma_k when wgrad_mfma_can_overwrite(g)
    int overwrite_bias;   // dbias = instead of += (pai_conv_wgrad_overwrite; 0 for pai_conv_wgrad_overwrite_w: the caller cleared dbias)
    float* slab;     // set by the launcher: pixel split `s` stores its tile into slab + s * |dW| (plain stores, no atomics)
    // prologue of x1 as in FwdArgs (pai_conv_wgrad_pro): gg_wgrad_mfma_k on pointwise layers, grouped3_wgrad_k
    const float *pscale, *pshift;
    int pact;
};
// the handle's weight-gradient workspace if it holds `bytes`, else NULL (the launch then adds with fp32 atomics)
float* wgrad_slab_acquire(int64_t bytes);
int launch_wgrad_slab_sum(float* dw, const float* slab, int nsplits, int64_t n, int overwrite, hipStream_t s);
// weight gradient of the same layer: diagonal 16-channel blocks only, partial blocks in the weight-gradient workspace
bool grouped3_wgrad_ok(int dtype, const GG& g, const float* dbias);
int64_t grouped3_wgrad_part_bytes(const GG& g);
int launch_grouped3_wgrad(const GG& g, const WgradArgs& a, float* part, hipStream_t s);
// patch-resident weight gradient with 128 x 64 / 64 x 128 wave tiles (gg_wg3.hip)
bool wgrad3_ok(const GG& g);
int launch_wgrad3(const GG& g, const WgradArgs& a, hipStream_t s);
const char* wgrad3_kernel_name(const GG& g);
int64_t wgrad3_slab_bytes(const GG& g);
bool wgrad3_overwrites(const GG& g);
// un-split, single-phase launch of gg_wgrad_mfma_k: every dW element has exactly one writer
bool wgrad_mfma_can_overwrite(const GG& g);
int launch_wgrad_simt(int dtype, const GG& g, const WgradArgs& a, hipStream_t s);
int launch_wgrad_rowdot(int dtype, const GG& g, const WgradArgs& a, hipStream_t s);
int launch_wgrad_mfma(const GG& g, const WgradArgs& a, hipStream_t s);
bool wgrad_mfma_ok(int dtype, const GG& g);
bool thin_wgrad_conv_ok(int dtype, const GG& g);
bool thin_wgrad_convt_ok(int dtype, const GG& g);
int launch_thin_wgrad_conv(const GG& g, const WgradArgs& a, hipStream_t s);
int launch_thin_wgrad_convt(const GG& g, const WgradArgs& a, hipStream_t s);
bool thin_wgrad_conv1_ok(int dtype, const GG& g);
int launch_thin_wgrad_conv1(const GG& g, const WgradArgs& a, hipStream_t s);
bool thin_wgrad_conv3_ok(int dtype, const GG& g);
int launch_thin_wgrad_conv3(const GG& g, const WgradArgs& a, hipStream_t s);
bool thin_wgrad_conv3t_ok(int dtype, const GG& g);
int launch_thin_wgrad_conv3t(const GG& g, const WgradArgs& a, hipStream_t s);
int64_t thin_wgrad_scratch_bytes(int64_t M, int T, int WC);

// Stand-alone micro-benchmark of the convolution family THROUGH THE C ABI (include/pai_hip.h): no torch, no Python.
// For every dense layer of BASELINE configs[1] (Pix2Pix generator + PatchGAN, batch 64, bf16) it times forward, input
// gradient and weight gradient under several tunable settings (pai_set_tunable), interleaved in one process (guide
// rule 24), and checks every setting's result against the first one -- on random data (tolerance) and on
// small-integer data, where every partial sum is exact in fp32 and the bf16 outputs must agree BIT FOR BIT whatever
// the summation order.
//
// build: hipcc -O2 --offload-arch=gfx950 scripts/micro/convbench.hip -Iinclude -Lthesis-pai-reconstruction_amd
//              -lpai_hip -Wl,-rpath,'$ORIGIN/../../thesis-pai-reconstruction_amd' -o scripts/micro/convbench
// usage: convbench [--filter name] [--ops fdw] [--iters N] [--rounds R] [--batch B] [--bias] [--bnbwd] [--zeros PCT] [--cold] [--set name=v,name=v ;...]
//        every --set adds one setting (comma-separated tunables); default: the library defaults only.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "pai_hip.h"

#define HCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)
#define PCHECK(x) do { int r_ = (x); if (r_) { fprintf(stderr, "%s:%d pai error %d: %s\n", __FILE__, __LINE__, r_, pai_last_error()); exit(3); } } while (0)

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ unsigned short f2bf_rne(float f) {
    unsigned u = __float_as_uint(f);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
// mode 0: uniform [-scale, scale); mode 1: integers in [-imax, imax]
// zero_pct: that share of the elements is exactly zero (post-ReLU activations of a trained network are ~half zeros: the
// chip draws less power on them and holds a higher clock than on dense random operands, guide rule 25)
__global__ void fill_bf16(unsigned short* p, size_t n, unsigned seed, float scale, int mode, int imax, int zero_pct = 0) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned h = hash32((unsigned)i * 2654435761u + seed);
        float v;
        if (mode == 0) v = ((h >> 8) * (1.0f / 8388608.0f) - 1.0f) * scale;
        else v = (float)((int)(h % (unsigned)(2 * imax + 1)) - imax);
        if (zero_pct && (int)(hash32(h ^ 0x9e3779b9u) % 100u) < zero_pct) v = 0.f;
        p[i] = f2bf_rne(v);
    }
}
__global__ void diff_bf16(const unsigned short* a, const unsigned short* b, size_t n, float* out /* [maxabs_a, maxdiff, nmismatch] */) {
    float ma = 0.f, md = 0.f, nm = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((unsigned)a[i] << 16), y = __uint_as_float((unsigned)b[i] << 16);
        ma = fmaxf(ma, fabsf(x));
        md = fmaxf(md, fabsf(x - y));
        if (a[i] != b[i]) nm += 1.f;
    }
    atomicMax((int*)&out[0], __float_as_int(ma));
    atomicMax((int*)&out[1], __float_as_int(md));
    atomicAdd(&out[2], nm);
}
__global__ void diff_f32(const float* a, const float* b, size_t n, float* out) {
    float ma = 0.f, md = 0.f, nm = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        ma = fmaxf(ma, fabsf(a[i]));
        md = fmaxf(md, fabsf(a[i] - b[i]));
        if (a[i] != b[i]) nm += 1.f;
    }
    atomicMax((int*)&out[0], __float_as_int(ma));
    atomicMax((int*)&out[1], __float_as_int(md));
    atomicAdd(&out[2], nm);
}

struct Layer { const char* name; int tr, H, C1, C2, Cout, nmul; };
static const Layer LAYERS[] = {
    {"enc1", 0, 128, 64, 0, 128, 1}, {"enc2", 0, 64, 128, 0, 256, 1}, {"enc3", 0, 32, 256, 0, 512, 1},
    {"enc4", 0, 16, 512, 0, 512, 1}, {"enc5", 0, 8, 512, 0, 512, 1}, {"enc6", 0, 4, 512, 0, 512, 1},
    {"enc7", 0, 2, 512, 0, 512, 1}, {"dec0", 1, 1, 512, 0, 512, 1}, {"dec1", 1, 2, 512, 512, 512, 1},
    {"dec2", 1, 4, 512, 512, 512, 1}, {"dec3", 1, 8, 512, 512, 512, 1}, {"dec4", 1, 16, 512, 512, 256, 1},
    {"dec5", 1, 32, 256, 256, 128, 1}, {"dec6", 1, 64, 128, 128, 64, 1},
    {"D1x2", 0, 128, 64, 0, 128, 2}, {"D2x2", 0, 64, 128, 0, 256, 2}, {"D3x2", 0, 32, 256, 0, 512, 2},
    // thin (HBM-bound) layers: encoders[0], discriminator block 0 (two 1-channel sources), decoders[7] (head)
    {"thin_enc0", 0, 256, 1, 0, 64, 1}, {"thin_D0x2", 0, 256, 1, 1, 64, 2}, {"thin_dec7", 1, 128, 64, 64, 1, 1}};

struct Setting { std::string label; std::vector<std::pair<std::string, int>> kv; };

static void apply_setting(const Setting& s) {
    // reset every tunable any setting mentions to "unset" is not possible; settings must name the same keys
    for (auto& kv : s.kv) PCHECK(pai_set_tunable(kv.first.c_str(), kv.second));
}

static std::vector<float> read3(float* d) {
    std::vector<float> h(3);
    HCHECK(hipMemcpy(h.data(), d, 12, hipMemcpyDeviceToHost));
    return h;
}

int main(int argc, char** argv) {
    const char* filter = "";
    const char* ops = "fdw";
    int iters = 10, rounds = 3, batch = 64, bias = 0, bnbwd = 0, zero_pct = 0, cold = 0;
    std::vector<Setting> settings;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--filter") && i + 1 < argc) filter = argv[++i];
        else if (!strcmp(argv[i], "--ops") && i + 1 < argc) ops = argv[++i];
        else if (!strcmp(argv[i], "--iters") && i + 1 < argc) iters = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--rounds") && i + 1 < argc) rounds = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--cold")) cold = 1;   // a 1 GB fill in front of every timed launch (one event pair each)
        else if (!strcmp(argv[i], "--batch") && i + 1 < argc) batch = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--bnbwd")) bnbwd = 1; // input gradients run pai_conv_dgrad_bn: producer BatchNorm / activation backward
                                                         // (affine pre-activation, second gradient, partial sums) fused into the store
        else if (!strcmp(argv[i], "--zeros") && i + 1 < argc) zero_pct = atoi(argv[++i]);   // percent of exact zeros in x1 / x2 / dy
        else if (!strcmp(argv[i], "--bias")) bias = 1;   // weight gradients also produce (and compare) the bias gradient
        else if (!strcmp(argv[i], "--set") && i + 1 < argc) {
            Setting s;
            s.label = argv[++i];
            char* dup = strdup(s.label.c_str());
            for (char* tok = strtok(dup, ","); tok; tok = strtok(nullptr, ",")) {
                char* eq = strchr(tok, '=');
                if (!eq) { fprintf(stderr, "bad --set %s\n", tok); return 1; }
                *eq = 0;
                s.kv.push_back({tok, atoi(eq + 1)});
            }
            free(dup);
            settings.push_back(s);
        } else { fprintf(stderr, "unknown argument %s\n", argv[i]); return 1; }
    }
    if (settings.empty()) settings.push_back(Setting{"default", {}});
    const int NS = (int)settings.size();

    int cus = 0, lds = 0;
    char arch[64];
    PCHECK(pai_device_info(&cus, &lds, arch, sizeof(arch)));
    printf("device %s, %d CUs; ABI version %d; batch %d, %d iters x %d rounds per setting\n", arch, cus, pai_version(), batch, iters, rounds);

    // split-K workspace / thin scratch: generous fixed sizes
    void *ws = nullptr, *scratch = nullptr;
    const int64_t ws_bytes = 512ll << 20, sc_bytes = 512ll << 20;
    HCHECK(hipMalloc(&ws, ws_bytes)); HCHECK(hipMemset(ws, 0, ws_bytes));
    HCHECK(hipMalloc(&scratch, sc_bytes));
    PCHECK(pai_set_workspace(ws, ws_bytes));
    PCHECK(pai_set_scratch(scratch, sc_bytes));
    void* wslab = nullptr;
    const int64_t wslab_bytes = 256ll << 20;    // weight-gradient slabs: 512 workgroup tiles of 128 KB = 64 MB per launch
    HCHECK(hipMalloc(&wslab, wslab_bytes));
    PCHECK(pai_set_wgrad_workspace(wslab, wslab_bytes));
    float* dstat;
    HCHECK(hipMalloc(&dstat, 12));
    hipStream_t st;
    HCHECK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    HCHECK(hipEventCreate(&e0)); HCHECK(hipEventCreate(&e1));
    void* junk = nullptr;
    if (cold) HCHECK(hipMalloc(&junk, 1ll << 30));

    std::vector<double> total(NS * 3, 0.0);
    int failures = 0;
    for (const Layer& L : LAYERS) {
        if (filter[0] && !strstr(L.name, filter)) continue;
        const int n = batch * L.nmul;
        pai_conv_desc d;
        memset(&d, 0, sizeof(d));
        d.dtype = PAI_BF16; d.transposed = L.tr; d.N = n; d.H = L.H; d.W = L.H; d.C1 = L.C1; d.C2 = L.C2; d.Cout = L.Cout;
        d.kernel = 4; d.stride = 2; d.pad = 1; d.relu1 = L.tr && L.Cout > 2; d.relu2 = (L.C2 && L.Cout > 2) ? L.tr : 0; d.epilogue_act = PAI_ACT_NONE;
        int OH, OW;
        PCHECK(pai_conv_out_hw(&d, &OH, &OW));
        const int Cin = L.C1 + L.C2;
        const size_t nx1 = (size_t)n * L.H * L.H * L.C1, nx2 = (size_t)n * L.H * L.H * L.C2, ny = (size_t)n * OH * OW * L.Cout;
        const size_t nw = (size_t)L.Cout * 16 * Cin;
        unsigned short *x1, *x2 = nullptr, *wf, *wd, *dy, *y[2], *dx1[2], *dx2[2] = {nullptr, nullptr};
        float *dw[2], *db[2], *stats, *bnp = nullptr, *bnpart = nullptr;
        unsigned short *bz = nullptr, *badd = nullptr;
        HCHECK(hipMalloc(&x1, nx1 * 2));
        if (nx2) HCHECK(hipMalloc(&x2, nx2 * 2));
        HCHECK(hipMalloc(&wf, nw * 4)); HCHECK(hipMalloc(&wd, nw * 4)); HCHECK(hipMalloc(&dy, ny * 2));
        for (int k = 0; k < 2; ++k) {
            HCHECK(hipMalloc(&y[k], ny * 2)); HCHECK(hipMalloc(&dx1[k], nx1 * 2));
            if (nx2) HCHECK(hipMalloc(&dx2[k], nx2 * 2));
            HCHECK(hipMalloc(&dw[k], nw * 4));
            HCHECK(hipMalloc(&db[k], (size_t)L.Cout * 4));
        }
        const int srows = pai_bn_stats_buffer_rows(pai_conv_fwd_stats_rows_max(&d));
        HCHECK(hipMalloc(&stats, (size_t)srows * 2 * L.Cout * 4));
        if (bnbwd && L.C1 > 2) {
            HCHECK(hipMalloc(&bz, nx1 * 2)); HCHECK(hipMalloc(&badd, nx1 * 2));
            HCHECK(hipMalloc(&bnp, (size_t)4 * L.C1 * 4));
            HCHECK(hipMalloc(&bnpart, (size_t)pai_conv_dgrad_bn_rows_max(&d) * 2 * L.C1 * 4));
            fill_bf16<<<1024, 256, 0, st>>>(bz, nx1, 21u, 1.0f, 1, 3);
            fill_bf16<<<1024, 256, 0, st>>>(badd, nx1, 22u, 1.0f, 1, 2);
            std::vector<float> hp(4 * L.C1);
            for (int c = 0; c < L.C1; ++c) { hp[c] = (c & 1) ? 1.f : 2.f; hp[L.C1 + c] = (float)((c % 5) - 2); hp[2 * L.C1 + c] = (float)(c % 3) - 1.f; hp[3 * L.C1 + c] = (c & 2) ? 0.5f : 1.f; }
            HCHECK(hipMemcpy(bnp, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
        }
        const double gflop = 2.0 * (double)n * L.H * L.H * (L.tr ? 1.0 : 0.25) * 16.0 * Cin * L.Cout / 1e9;

        auto fill_all = [&](int mode) {
            // integer mode: |x| <= 2, |w| <= 2, K <= 16384 terms -> |sum| <= 65536 < 2^24: exact in fp32
            fill_bf16<<<1024, 256, 0, st>>>(x1, nx1, 11u, 1.0f, mode, 2, zero_pct);
            if (nx2) fill_bf16<<<1024, 256, 0, st>>>(x2, nx2, 12u, 1.0f, mode, 2, zero_pct);
            fill_bf16<<<1024, 256, 0, st>>>(wf, nw, 13u, 0.05f, mode, 2);
            fill_bf16<<<1024, 256, 0, st>>>(wd, nw, 14u, 0.05f, mode, 2);
            fill_bf16<<<1024, 256, 0, st>>>(dy, ny, 15u, 1.0f, mode, 2, zero_pct);
            HCHECK(hipStreamSynchronize(st));
        };
        auto run = [&](char op, int k) {
            if (op == 'f') PCHECK(pai_conv_fwd(&d, x1, x2, wf, nullptr, y[k], nullptr, nullptr, (L.C1 > 2 && L.Cout > 2) ? stats : nullptr, st));
            else if (op == 'd' && bz) {
                pai_bwd_epilogue e;
                e.z = bz; e.add = L.tr ? nullptr : badd; e.scale = bnp; e.shift = bnp + L.C1; e.mean = bnp + 2 * L.C1; e.rstd = bnp + 3 * L.C1;
                e.partials = bnpart; e.act1 = L.tr ? PAI_ACT_RELU : PAI_ACT_LRELU; e.act2 = L.tr ? PAI_ACT_NONE : PAI_ACT_RELU;
                // the thin head (decoders[7], one output channel): the engine's call -- decoders[6] is read WITHOUT an activation, so
                // du IS the gradient stored: no scale / shift, act1 = none (thin_fwd2_k<..., BWD>; round 6: the ReLU form above took
                // the two-launch path and never reached that kernel)
                if (L.tr && L.Cout <= 2) { e.scale = nullptr; e.shift = nullptr; e.act1 = PAI_ACT_NONE; }
                int rows = 0;
                PCHECK(pai_conv_dgrad_bn(&d, dy, wd, dx1[k], dx2[k], &e, &rows, st));
            } else if (op == 'd') PCHECK(pai_conv_dgrad(&d, dy, wd, dx1[k], dx2[k], 0, st));
            else {
                HCHECK(hipMemsetAsync(dw[k], 0, nw * 4, st));
                HCHECK(hipMemsetAsync(db[k], 0, (size_t)L.Cout * 4, st));
                PCHECK(pai_conv_wgrad(&d, x1, x2, dy, dw[k], bias ? db[k] : nullptr, st));
            }
        };
        auto compare = [&](char op, const char* what, bool exact, const char* label) {
            HCHECK(hipMemsetAsync(dstat, 0, 12, st));
            if (op == 'f') diff_bf16<<<1024, 256, 0, st>>>(y[0], y[1], ny, dstat);
            else if (op == 'd') {
                diff_bf16<<<1024, 256, 0, st>>>(dx1[0], dx1[1], nx1, dstat);
                if (nx2) diff_bf16<<<1024, 256, 0, st>>>(dx2[0], dx2[1], nx2, dstat);
            } else {
                diff_f32<<<1024, 256, 0, st>>>(dw[0], dw[1], nw, dstat);
                if (bias) diff_f32<<<64, 256, 0, st>>>(db[0], db[1], (size_t)L.Cout, dstat);
            }
            HCHECK(hipStreamSynchronize(st));
            const auto r = read3(dstat);
            const bool ok = exact ? (r[2] == 0.f) : (r[1] <= 0.02f * r[0] + 1e-6f);
            if (!ok || r[0] == 0.f) {
                ++failures;
                printf("  MISMATCH %s %c [%s] %s: max|ref| %.5g max|diff| %.5g mismatching elements %.0f\n", L.name, op, label, what, r[0], r[1], r[2]);
            }
        };

        char kname[3][128] = {"", "", ""};
        for (int oi = 0; oi < 3; ++oi) {
            const char op = "fdw"[oi];
            if (!strchr(ops, op)) continue;
            // ---- correctness of every setting against setting 0: exact on integer data, tolerance on random data
            for (int mode = 1; mode >= 0; --mode) {
                fill_all(mode);
                apply_setting(settings[0]);
                run(op, 0);
                for (int s = 1; s < NS; ++s) {
                    apply_setting(settings[s]);
                    run(op, 1);
                    compare(op, mode ? "integer data (must be bit-exact)" : "random data", mode == 1, settings[s].label.c_str());
                }
            }
            // ---- timing: interleaved rounds, random data
            std::vector<std::vector<float>> us(NS);
            for (int r = 0; r < rounds + 1; ++r)
                for (int s = 0; s < NS; ++s) {
                    apply_setting(settings[s]);
                    if (r == 0) pai_conv_kernel_name(&d, oi, kname[oi], sizeof(kname[oi]));
                    run(op, 0);   // warm
                    float ms;
                    if (cold) {
                        // every tensor of a training step is cold (the memory-side cache holds 256 MB): relaunching on the
                        // same buffers flatters the HBM-bound layers
                        float sum = 0.f;
                        for (int it = 0; it < iters; ++it) {
                            HCHECK(hipMemsetAsync(junk, it, 1ll << 30, st));
                            HCHECK(hipEventRecord(e0, st));
                            if (op == 'w') PCHECK(pai_conv_wgrad(&d, x1, x2, dy, dw[0], bias ? db[0] : nullptr, st));
                            else run(op, 0);
                            HCHECK(hipEventRecord(e1, st));
                            HCHECK(hipEventSynchronize(e1));
                            HCHECK(hipEventElapsedTime(&ms, e0, e1));
                            sum += ms;
                        }
                        ms = sum;
                    } else {
                        HCHECK(hipEventRecord(e0, st));
                        for (int it = 0; it < iters; ++it) {
                            if (op == 'w') PCHECK(pai_conv_wgrad(&d, x1, x2, dy, dw[0], bias ? db[0] : nullptr, st));
                            else run(op, 0);
                        }
                        HCHECK(hipEventRecord(e1, st));
                        HCHECK(hipEventSynchronize(e1));
                        HCHECK(hipEventElapsedTime(&ms, e0, e1));
                    }
                    if (r > 0) us[s].push_back(ms * 1e3f / iters);
                }
            printf("%-5s %c %7.1f GF |", L.name, op, gflop);
            for (int s = 0; s < NS; ++s) {
                std::sort(us[s].begin(), us[s].end());
                const float med = us[s][us[s].size() / 2], mn = us[s][0];
                total[s * 3 + oi] += med;
                printf(" [%s] med %7.1f us %5.0f TF (min %7.1f)", settings[s].label.c_str(), med, gflop * 1e3 / med, mn);
            }
            apply_setting(settings[NS - 1]);
            pai_conv_kernel_name(&d, oi, kname[oi], sizeof(kname[oi]));
            printf(" | last: %s\n", kname[oi]);
            fflush(stdout);
        }
        if (bz) { (void)hipFree(bz); (void)hipFree(badd); (void)hipFree(bnp); (void)hipFree(bnpart); }
        (void)hipFree(x1); if (x2) (void)hipFree(x2); (void)hipFree(wf); (void)hipFree(wd); (void)hipFree(dy); (void)hipFree(stats);
        for (int k = 0; k < 2; ++k) { (void)hipFree(y[k]); (void)hipFree(dx1[k]); if (dx2[k]) (void)hipFree(dx2[k]); (void)hipFree(dw[k]); (void)hipFree(db[k]); }
    }
    for (int s = 0; s < NS; ++s)
        printf("total [%s]: fwd %.1f us, dgrad %.1f us, wgrad %.1f us\n", settings[s].label.c_str(), total[s * 3], total[s * 3 + 1], total[s * 3 + 2]);
    printf("%s\n", failures ? "FAILURES" : "all settings agree");
    return failures ? 1 : 0;
}

// C-ABI entry points of libpai_hip.so: argument checking, problem construction and
// kernel selection for the convolution family (include/pai_hip.h).
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

void pai_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* pai_last_error(void) { return g_err; }
extern "C" int pai_version(void) { return 132; }   // 110: handles, tunables, device-side Adam step, *_take, pack multi; 120: weight-gradient workspace; 121: pai_adam_pack, pai_bn_bwd_apply_affine; 130: launch plans; 131: pai_lerp_multi; 132: input prologue (pai_conv_fwd_pro / pai_conv_wgrad_pro)

// build-option bits; none since ABI 130 (bit 0 announced the round-2 experiment kernels, which were removed)
extern "C" int pai_build_flags(void) { return 0; }

extern "C" int pai_device_info(int* cu_count, int* lds_bytes, char* arch_name, int arch_name_len) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    PAI_CHECK(e == hipSuccess, "hipGetDevice: %s", hipGetErrorString(e));
    hipDeviceProp_t p;
    e = hipGetDeviceProperties(&p, dev);
    PAI_CHECK(e == hipSuccess, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (lds_bytes) *lds_bytes = (int)p.sharedMemPerBlock;
    if (arch_name && arch_name_len > 0) {
        strncpy(arch_name, p.gcnArchName, arch_name_len - 1);
        arch_name[arch_name_len - 1] = 0;
    }
    return 0;
}

// ---- run-time tunables ------------------------------------------------------------------------------
// A small name -> value table (host side, process-wide): kernel-selection switches that tests use to pin the kernel
// family a call runs and that the micro-benchmarks flip to time two kernels in one process.
struct Tunable { char name[32]; int value; };
static Tunable g_tunables[32];
static int g_ntunables = 0;

// Default of a tunable from the environment: PAI_TUNE_<name>=<int> (looked up once per name).  For whole-suite A/B runs
// (pytest, bench.py) of a kernel-selection switch without touching the callers; pai_set_tunable still wins.
struct TunableEnv { char name[32]; int has, value; };
static TunableEnv g_tunable_env[48];
static int g_ntunable_env = 0;
static bool tunable_env(const char* name, int* value) {
    for (int i = 0; i < g_ntunable_env; ++i)
        if (!strcmp(g_tunable_env[i].name, name)) { *value = g_tunable_env[i].value; return g_tunable_env[i].has != 0; }
    char key[64];
    snprintf(key, sizeof(key), "PAI_TUNE_%s", name);
    const char* e = getenv(key);
    if (g_ntunable_env < 48 && strlen(name) < sizeof(g_tunable_env[0].name)) {
        TunableEnv& t = g_tunable_env[g_ntunable_env];
        strcpy(t.name, name);
        t.has = e != nullptr;
        t.value = e ? atoi(e) : 0;
        ++g_ntunable_env;   // (published last: a racing reader sees either no entry or a complete one)
    }
    *value = e ? atoi(e) : 0;
    return e != nullptr;
}

int pai_tunable(const char* name, int def) {
    for (int i = 0; i < g_ntunables; ++i)
        if (!strcmp(g_tunables[i].name, name)) return g_tunables[i].value;
    int v;
    return tunable_env(name, &v) ? v : def;
}

extern "C" int pai_set_tunable(const char* name, int value) {
    PAI_CHECK(name && strlen(name) < sizeof(g_tunables[0].name), "pai_set_tunable: bad name");
    for (int i = 0; i < g_ntunables; ++i)
        if (!strcmp(g_tunables[i].name, name)) {
            if (value == PAI_TUNABLE_UNSET) g_tunables[i] = g_tunables[--g_ntunables];   // back to the built-in default
            else g_tunables[i].value = value;
            return 0;
        }
    if (value == PAI_TUNABLE_UNSET) return 0;
    PAI_CHECK(g_ntunables < 32, "pai_set_tunable: table full");
    strcpy(g_tunables[g_ntunables].name, name);
    g_tunables[g_ntunables++].value = value;
    return 0;
}

static void finish_gg(GG* g);

// ---------------------------------------------------------------------------------
static int check_desc(const pai_conv_desc* d) {
    PAI_CHECK(d != nullptr, "null descriptor");
    PAI_CHECK(d->dtype == PAI_F32 || d->dtype == PAI_BF16, "bad dtype %d", d->dtype);
    if (d->kernel == 1) {   // pointwise conv of the attention gates (models/attention_unet.py:72-84)
        PAI_CHECK(d->pad == 0 && d->stride == 1 && !d->transposed, "kernel=1 needs pad=0 stride=1 Conv2d");
    } else if (d->kernel == 3) {   // 3x3 "same" conv of the residual U-Net (models/res_unet.py:59,62,90,117,265,308)
        PAI_CHECK(d->pad == 1 && d->stride == 1 && !d->transposed, "kernel=3 needs pad=1 stride=1 Conv2d");
    } else {
        PAI_CHECK(d->kernel == 4 && d->pad == 1, "only kernel=4 pad=1 or kernel=1 pad=0 supported (got k=%d p=%d)",
                  d->kernel, d->pad);
    }
    PAI_CHECK(d->N > 0 && d->H > 0 && d->W > 0 && d->C1 > 0 && d->C2 >= 0 && d->Cout > 0,
              "bad shape N=%d H=%d W=%d C1=%d C2=%d Cout=%d", d->N, d->H, d->W, d->C1, d->C2,
              d->Cout);
    if (d->transposed) {
        PAI_CHECK(d->stride == 2, "ConvTranspose2d needs stride 2");
    } else {
        PAI_CHECK(d->stride == 1 || d->stride == 2, "Conv2d stride must be 1 or 2");
        if (d->stride == 2)
            PAI_CHECK((d->H % 2) == 0 && (d->W % 2) == 0, "stride-2 Conv2d needs even H, W");
        else if (d->kernel == 4)
            PAI_CHECK(d->H >= 2 && d->W >= 2, "k4 s1 p1 Conv2d needs H, W >= 2");
    }
    PAI_CHECK((int64_t)d->N * d->H * d->W * 4 < (int64_t)1 << 31, "problem too large for int32 rows");
    PAI_CHECK(d->pack_flags == 0 && (d->hints & ~PAI_HINT_SOLO) == 0, "pai_conv_desc: pack_flags / unknown hint bits must be zero (got %d, %d)",
              d->pack_flags, d->hints);
    if (d->groups > 1) {
        PAI_CHECK(d->kernel == 3 && d->C2 == 0 && d->C1 == d->Cout && (d->C1 % d->groups) == 0 && (d->C1 % 16) == 0 &&
                      (16 % (d->C1 / d->groups)) == 0,
                  "groups=%d needs a 3x3 Conv2d with C1 = Cout = 16 k and groups that tile 16-channel slices", d->groups);
    }
    return 0;
}

extern "C" int pai_conv_out_hw(const pai_conv_desc* d, int* OH, int* OW) {
    if (check_desc(d)) return 1;
    if (d->transposed) {
        *OH = d->H * 2;
        *OW = d->W * 2;
    } else {
        *OH = (d->H + 2 * d->pad - d->kernel) / d->stride + 1;
        *OW = (d->W + 2 * d->pad - d->kernel) / d->stride + 1;
    }
    return 0;
}

// phase decomposition of a k4 s2 p1 transposed gather: output index o = 2a + ph receives
// taps kh with kh == ph+1 (mod 2), from source index a + off.
static void phase_taps(int ph, int kh[2], int off[2]) {
    if (ph == 0) {
        kh[0] = 1; off[0] = 0;
        kh[1] = 3; off[1] = -1;
    } else {
        kh[0] = 0; off[0] = 1;
        kh[1] = 2; off[1] = 0;
    }
}

static void fill_conv_taps(GG* g, int S, int off0) {
    // stride-S conv form: source = grid*S + (kh + off0)
    g->S = S;
    g->nphase = 1;
    g->ntaps = 16;
    g->OS = 1;
    g->poy[0] = g->pox[0] = 0;
    for (int kh = 0; kh < 4; ++kh)
        for (int kw = 0; kw < 4; ++kw) {
            int t = kh * 4 + kw;
            g->dy[0][t] = (signed char)(kh + off0);
            g->dx[0][t] = (signed char)(kw + off0);
            g->wt[0][t] = (signed char)t;
        }
}

static void fill_pointwise(GG* g) {
    g->S = 1; g->nphase = 1; g->ntaps = 1; g->OS = 1; g->wtaps = 1;
    g->poy[0] = g->pox[0] = 0;
    g->dy[0][0] = g->dx[0][0] = g->wt[0][0] = 0;
}

// k3 s1 p1: out[i] = sum_kh in[i + kh - 1] w[kh]  (sign = +1);  its input gradient dx[i] = sum_kh dy[i + 1 - kh] w[kh]
static void fill_3x3(GG* g, int sign) {
    g->S = 1; g->nphase = 1; g->ntaps = 9; g->OS = 1; g->wtaps = 9;
    g->poy[0] = g->pox[0] = 0;
    for (int kh = 0; kh < 3; ++kh)
        for (int kw = 0; kw < 3; ++kw) {
            const int t = kh * 3 + kw;
            g->dy[0][t] = (signed char)(sign * (kh - 1));
            g->dx[0][t] = (signed char)(sign * (kw - 1));
            g->wt[0][t] = (signed char)t;
        }
}

static void fill_phase_taps(GG* g) {
    g->S = 1;
    g->nphase = 4;
    g->ntaps = 4;
    g->OS = 2;
    for (int ph = 0; ph < 2; ++ph)
        for (int pw = 0; pw < 2; ++pw) {
            int p = ph * 2 + pw;
            g->poy[p] = (signed char)ph;
            g->pox[p] = (signed char)pw;
            int khs[2], offy[2], kws[2], offx[2];
            phase_taps(ph, khs, offy);
            phase_taps(pw, kws, offx);
            for (int a = 0; a < 2; ++a)
                for (int b = 0; b < 2; ++b) {
                    int t = a * 2 + b;
                    g->dy[p][t] = (signed char)offy[a];
                    g->dx[p][t] = (signed char)offx[b];
                    g->wt[p][t] = (signed char)(khs[a] * 4 + kws[b]);
                }
        }
}

int gg_build_fwd(const pai_conv_desc* d, GG* g) {
    if (check_desc(d)) return 1;
    memset(g, 0, sizeof(*g));
    g->N = d->N; g->H = d->H; g->W = d->W;
    g->C1 = d->C1; g->C2 = d->C2; g->Cin = d->C1 + d->C2;
    g->Cout = d->Cout;
    g->D1 = d->Cout; g->D2 = 0;
    g->wtaps = 16;
    g->gslice = d->groups > 1 ? 16 : 0;
    g->solo = (d->hints & PAI_HINT_SOLO) ? 1 : 0;
    g->relu1 = d->relu1; g->relu2 = d->relu2;
    int OH, OW;
    pai_conv_out_hw(d, &OH, &OW);
    g->OH = OH; g->OW = OW;
    if (d->kernel == 1) {
        g->OHg = OH; g->OWg = OW;
        fill_pointwise(g);
    } else if (d->kernel == 3) {
        g->OHg = OH; g->OWg = OW;
        fill_3x3(g, +1);
    } else if (!d->transposed) {
        g->OHg = OH; g->OWg = OW;
        fill_conv_taps(g, d->stride, -d->pad);
    } else {
        // out[2a+ph] gathers in[a + off]  (models/pix2pix.py:99-105 semantics)
        g->OHg = d->H; g->OWg = d->W;
        fill_phase_taps(g);
    }
    finish_gg(g);
    return 0;
}

int gg_build_dgrad(const pai_conv_desc* d, GG* g) {
    if (check_desc(d)) return 1;
    memset(g, 0, sizeof(*g));
    int OH, OW;
    pai_conv_out_hw(d, &OH, &OW);
    // source = dy [N, OH, OW, Cout];  destination = dx [N, H, W, C1|C2]
    g->N = d->N; g->H = OH; g->W = OW;
    g->C1 = d->Cout; g->C2 = 0; g->Cin = d->Cout;
    g->Cout = d->C1 + d->C2;
    g->D1 = d->C1; g->D2 = d->C2;
    g->wtaps = 16;
    g->gslice = d->groups > 1 ? 16 : 0;
    g->OH = d->H; g->OW = d->W;
    if (d->kernel == 1) {
        g->OHg = d->H; g->OWg = d->W;
        fill_pointwise(g);    // dx = dy x W^T, pixel by pixel
    } else if (d->kernel == 3) {
        g->OHg = d->H; g->OWg = d->W;
        fill_3x3(g, -1);
    } else if (!d->transposed) {
        if (d->stride == 2) {
            // dx[2a+ph] = sum_kh dy[a + off] w[kh]  -- same phase structure as ConvTranspose2d
            g->OHg = d->H / 2; g->OWg = d->W / 2;
            fill_phase_taps(g);
        } else {
            // dx[i] = sum_kh dy[i + pad - kh] w[kh]
            g->OHg = d->H; g->OWg = d->W;
            g->S = 1; g->nphase = 1; g->ntaps = 16; g->OS = 1;
            for (int kh = 0; kh < 4; ++kh)
                for (int kw = 0; kw < 4; ++kw) {
                    int t = kh * 4 + kw;
                    g->dy[0][t] = (signed char)(d->pad - kh);
                    g->dx[0][t] = (signed char)(d->pad - kw);
                    g->wt[0][t] = (signed char)t;
                }
        }
    } else {
        // d in[i] = sum_kh d out[2i - 1 + kh] w[kh]  -- stride-2 conv form over d out
        g->OHg = d->H; g->OWg = d->W;
        fill_conv_taps(g, 2, -d->pad);
    }
    finish_gg(g);
    return 0;
}

static void finish_gg(GG* g) {
    g->M = g->N * g->OHg * g->OWg;
    auto lg = [](int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; };
    g->lw = lg(g->OWg);
    g->lh = lg(g->OHg);
    if (g->lw < 0 || g->lh < 0) g->lw = g->lh = -1;
    g->lsw = lg(g->W); g->lsh = lg(g->H); g->ldw = lg(g->OW); g->ldh = lg(g->OH);
    if (g->lw < 0 || g->lsw < 0 || g->lsh < 0 || g->ldw < 0 || g->ldh < 0)
        g->lsw = g->lsh = g->ldw = g->ldh = -1;
}

// ---- per-device handles ---------------------------------------------------------------------------------
// "no hidden global state except a per-device handle".  A handle owns the caller-provided split-K
// workspace and scratch of ONE device; several handles may exist per device (e.g. one per model replica), one of them
// is ACTIVE per device at a time (pai_bind) and serves the launches issued while that device is current.  The table
// below only maps device -> active handle.  Threading contract: one host thread drives a device at a time.
constexpr int PAI_MAX_DEVICES = 64;
static pai_handle_s* g_active[PAI_MAX_DEVICES];
static const pai_handle_s g_empty_handle = {-1, nullptr, 0, nullptr, 0, nullptr, 0};

static int current_device() {
    int dev = 0;
    return hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < PAI_MAX_DEVICES ? dev : -1;
}

const pai_handle_s* pai_ctx() {
    const int dev = current_device();
    return dev >= 0 && g_active[dev] ? g_active[dev] : &g_empty_handle;
}

extern "C" int pai_create(int device_id, void** handle_out) {
    PAI_CHECK(handle_out != nullptr, "pai_create: null handle_out");
    PAI_CHECK(device_id >= 0 && device_id < PAI_MAX_DEVICES, "pai_create: device id %d out of range", device_id);
    pai_handle_s* h = new pai_handle_s{device_id, nullptr, 0, nullptr, 0, nullptr, 0};
    if (!g_active[device_id]) g_active[device_id] = h;      // the first handle of a device is active at once
    *handle_out = h;
    return 0;
}

extern "C" int pai_bind(void* handle) {
    PAI_CHECK(handle != nullptr, "pai_bind: null handle");
    pai_handle_s* h = (pai_handle_s*)handle;
    g_active[h->device] = h;
    return 0;
}

extern "C" int pai_destroy(void* handle) {
    PAI_CHECK(handle != nullptr, "pai_destroy: null handle");
    pai_handle_s* h = (pai_handle_s*)handle;
    if (g_active[h->device] == h) g_active[h->device] = nullptr;
    delete h;          // the buffers are the caller's
    return 0;
}

extern "C" int pai_handle_set_workspace(void* handle, void* zeroed_device_memory, int64_t bytes) {
    PAI_CHECK(handle != nullptr, "pai_handle_set_workspace: null handle");
    pai_handle_s* h = (pai_handle_s*)handle;
    h->workspace = (float*)zeroed_device_memory;
    h->workspace_bytes = zeroed_device_memory ? bytes : 0;
    return 0;
}

extern "C" int pai_handle_set_scratch(void* handle, void* device_memory, int64_t bytes) {
    PAI_CHECK(handle != nullptr, "pai_handle_set_scratch: null handle");
    pai_handle_s* h = (pai_handle_s*)handle;
    h->scratch = (float*)device_memory;
    h->scratch_bytes = device_memory ? bytes : 0;
    return 0;
}

extern "C" int pai_handle_set_wgrad_workspace(void* handle, void* device_memory, int64_t bytes) {
    PAI_CHECK(handle != nullptr, "pai_handle_set_wgrad_workspace: null handle");
    pai_handle_s* h = (pai_handle_s*)handle;
    h->wslab = (float*)device_memory;
    h->wslab_bytes = device_memory ? bytes : 0;
    return 0;
}

float* wgrad_slab_acquire(int64_t bytes) {
    const pai_handle_s* h = pai_ctx();
    return (h->wslab && h->wslab_bytes >= bytes) ? h->wslab : nullptr;
}

// The active handle of the current device, created on first use: convenience for single-model callers
// (pai_set_workspace / pai_set_scratch without an explicit pai_create).
static pai_handle_s* active_or_new() {
    const int dev = current_device();
    if (dev < 0) return nullptr;
    if (!g_active[dev]) g_active[dev] = new pai_handle_s{dev, nullptr, 0, nullptr, 0, nullptr, 0};
    return g_active[dev];
}

extern "C" int pai_set_workspace(void* zeroed_device_memory, int64_t bytes) {
    pai_handle_s* h = active_or_new();
    PAI_CHECK(h != nullptr, "pai_set_workspace: no current HIP device");
    return pai_handle_set_workspace(h, zeroed_device_memory, bytes);
}

extern "C" int pai_set_scratch(void* device_memory, int64_t bytes) {
    pai_handle_s* h = active_or_new();
    PAI_CHECK(h != nullptr, "pai_set_scratch: no current HIP device");
    return pai_handle_set_scratch(h, device_memory, bytes);
}

extern "C" int pai_set_wgrad_workspace(void* device_memory, int64_t bytes) {
    pai_handle_s* h = active_or_new();
    PAI_CHECK(h != nullptr, "pai_set_wgrad_workspace: no current HIP device");
    return pai_handle_set_wgrad_workspace(h, device_memory, bytes);
}

extern "C" int64_t pai_conv_wgrad_workspace_bytes(const pai_conv_desc* d) {
    GG g;
    if (gg_build_fwd(d, &g)) return -1;
    if (grouped3_wgrad_ok(d->dtype, g, nullptr)) return grouped3_wgrad_part_bytes(g);
    if (!wgrad_mfma_ok(d->dtype, g)) return 0;
    return wgrad3_slab_bytes(g);
}

extern "C" int64_t pai_conv_workspace_bytes(const pai_conv_desc* d, int op) {
    GG g;
    if (op == 1 ? gg_build_dgrad(d, &g) : gg_build_fwd(d, &g)) return -1;
    if (op == 2) return 0;
    FwdArgs a;
    memset(&a, 0, sizeof(a));
    a.y1 = (void*)1;
    if (op == 1) a.y2 = (void*)1;
    if (!fwd_mfma_ok(d->dtype, g, a)) return 0;
    return fwd_mfma_workspace_bytes(g);
}

extern "C" int64_t pai_conv_scratch_bytes(const pai_conv_desc* d, int op) {
    GG g;
    if (op == 1 ? gg_build_dgrad(d, &g) : gg_build_fwd(d, &g)) return -1;
    if (op == 2) {
        if (thin_wgrad_conv_ok(d->dtype, g)) return thin_wgrad_scratch_bytes((int64_t)g.N * g.OHg * g.OWg, g.C1 + g.C2, g.Cout);
        if (thin_wgrad_convt_ok(d->dtype, g) || thin_wgrad_conv1_ok(d->dtype, g) || thin_wgrad_conv3t_ok(d->dtype, g))
            return thin_wgrad_scratch_bytes((int64_t)g.N * g.H * g.W, 1, g.Cin);
        if (thin_wgrad_conv3_ok(d->dtype, g)) return thin_wgrad_scratch_bytes((int64_t)g.N * g.OHg * g.OWg, 1, g.Cout);
        return 0;
    }
    FwdArgs a;
    memset(&a, 0, sizeof(a));
    if (thin_dgrad_shape_ok(d->dtype, g)) return thin_dgrad_scratch_bytes(g, a);
    return 0;
}

// ---------------------------------------------------------------------------------
static bool use_mfma(int dtype, const GG& g, const FwdArgs& a) { return fwd_mfma_ok(dtype, g, a); }

extern "C" int pai_conv_fwd_stats_rows(const pai_conv_desc* d) {
    GG g;
    if (gg_build_fwd(d, &g)) return -1;
    FwdArgs a;
    memset(&a, 0, sizeof(a));
    a.y1 = (void*)1;  // raw output present
    a.stats = (float*)1;
    if (grouped3_ok(d->dtype, g, a)) return grouped3_rows(g);
    if (pw_ok(d->dtype, g, a)) return pw_rows(g);
    if (pwx_ok(d->dtype, g, a)) return pwx_rows(g);
    if (!thin_fwd_ok(d->dtype, g, a) && !fwd_rowdot_ok(g, a) && !use_mfma(d->dtype, g, a) && small_ok(d->dtype, g, a))
        return small_rows(g);
    int mt = use_mfma(d->dtype, g, a) ? fwd_mfma_mtiles(g) : fwd_simt_mtiles(g);
    return mt * g.nphase;
}

extern "C" int pai_conv_kernel_id(const pai_conv_desc* d, int op) {
    GG g;
    FwdArgs a;
    memset(&a, 0, sizeof(a));
    a.y1 = (void*)1;
    if (op == 1) {
        if (gg_build_dgrad(d, &g)) return -1;
        a.y2 = (void*)1;
    } else {
        if (gg_build_fwd(d, &g)) return -1;
    }
    if (op == 2) {
        if (grouped3_wgrad_ok(d->dtype, g, nullptr) && wgrad_slab_acquire(grouped3_wgrad_part_bytes(g))) return 6;
        if (thin_wgrad_conv_ok(d->dtype, g) || thin_wgrad_convt_ok(d->dtype, g) || thin_wgrad_conv1_ok(d->dtype, g) ||
            thin_wgrad_conv3_ok(d->dtype, g) || thin_wgrad_conv3t_ok(d->dtype, g)) return 4;
        if (g.Cout <= 2 && (g.ntaps == 4 || g.ntaps == 9 || g.ntaps == 16) && (g.C1 % 8) == 0 && (g.C2 % 8) == 0) {
            int chunks = g.Cin / 8;
            if (chunks <= 256 && (chunks & (chunks - 1)) == 0) return 1;
        }
        if (wgrad_mfma_ok(d->dtype, g)) return (g.Cout % 128) == 0 ? 2 : 3;
        return 0;
    }
    if (op != 2 && grouped3_ok(d->dtype, g, a)) return 6;
    if (thin_fwd_ok(d->dtype, g, a) || thin_dgrad_ok(d->dtype, g, a) || pw_ok(d->dtype, g, a)) return 4;
    if (op == 1 && g.D2 == 0) {          // single-destination input gradients: the PatchGAN head's own kernel
        FwdArgs a1 = a;
        a1.y2 = nullptr;
        if (head_dgrad_ok(d->dtype, g, a1)) return 4;
    }
    if (fwd_rowdot_ok(g, a)) return 1;
    if (pwx_ok(d->dtype, g, a)) return 7;
    if (fwd_mfma_ok(d->dtype, g, a))
        return ((g.Cout % 128) == 0 && (g.D2 == 0 || (g.D1 % 128) == 0)) ? 2 : 3;
    if (small_ok(d->dtype, g, a)) return 5;
    return 0;
}

extern "C" int pai_conv_kernel_name(const pai_conv_desc* d, int op, char* name, int name_len) {
    static const char* fam[8] = {"gg_simt", "gg_rowdot", "gg_mfma", "gg_mfma", "thin_mfma_bf16", "small_mfma_bf16", "grouped3_k",
                                 "pwx_k"};
    const int id = pai_conv_kernel_id(d, op);
    if (id < 0 || !name || name_len <= 0) return -1;
    const char* n = (id == 6 && op == 2) ? "grouped3_wgrad_k" : fam[id];
    if (id == 2 || id == 3) {
        GG g;
        if (op == 1 ? gg_build_dgrad(d, &g) : gg_build_fwd(d, &g)) return -1;
        n = op == 2 ? wgrad_mfma_kernel_name(g) : fwd_mfma_kernel_name(g);
    }
    if (id == 7) {
        GG g;
        if (op == 1 ? gg_build_dgrad(d, &g) : gg_build_fwd(d, &g)) return -1;
        n = pwx_kernel_name(g);
    }
    strncpy(name, n, name_len - 1);
    name[name_len - 1] = 0;
    return 0;
}

// upper bound of pai_conv_fwd_stats_rows over every launch configuration the library may pick
// (the split-K path writes one row per 16 output rows); size the statistics buffer with this
extern "C" int pai_conv_fwd_stats_rows_max(const pai_conv_desc* d) {
    GG g;
    if (gg_build_fwd(d, &g)) return -1;
    return cdiv(g.M, 16) * g.nphase;
}

static int run_fwd(int dtype, const GG& g, const FwdArgs& a, hipStream_t s) {
    if (a.pscale) {                     // prologue on x1: only the kernels that apply it on load
        if (grouped3_ok(dtype, g, a)) return launch_grouped3(g, a, s);
        if (pwx_ok(dtype, g, a)) return launch_pwx(g, a, s);
        pai_set_error("pai_conv_fwd_pro: this layer takes no prologue (ask pai_conv_prologue_ok)");
        return 1;
    }
    if (grouped3_ok(dtype, g, a)) return launch_grouped3(g, a, s);   // grouped 3x3: 16-channel slices, patch in LDS
    if (pw_ok(dtype, g, a)) return launch_pw(g, a, s);
    if (thin_fwd_ok(dtype, g, a)) return launch_thin_fwd(g, a, s);
    if (head_dgrad_ok(dtype, g, a)) return launch_head_dgrad(g, a, s);
    if (thin_dgrad_ok(dtype, g, a)) return launch_thin_dgrad(g, a, s);
    if (fwd_rowdot_ok(g, a)) return launch_fwd_rowdot(dtype, g, a, s);
    if (pwx_ok(dtype, g, a)) return launch_pwx(g, a, s);             // big pointwise layers: streaming kernel
    if (use_mfma(dtype, g, a)) return launch_fwd_mfma(g, a, s);
    if (small_ok(dtype, g, a)) return launch_small(g, a, s);
    return launch_fwd_simt(dtype, g, a, s);
}

extern "C" int pai_conv_fwd(const pai_conv_desc* d, const void* x1, const void* x2,
                            const void* w_fwd, const float* bias, void* y_raw, void* y_act,
                            float* y_f32, float* stats, void* stream) {
    GG g;
    if (gg_build_fwd(d, &g)) return 1;
    PAI_CHECK(x1 && w_fwd, "pai_conv_fwd: null x1 / w");
    PAI_CHECK(d->C2 == 0 || x2, "pai_conv_fwd: C2 > 0 but x2 is null");
    PAI_CHECK(y_raw || y_act || y_f32, "pai_conv_fwd: no output requested");
    FwdArgs a;
    memset(&a, 0, sizeof(a));
    a.x1 = x1; a.x2 = x2; a.w = w_fwd; a.bias = bias;
    a.y1 = y_raw; a.y2 = nullptr; a.yact = y_act; a.yf32 = y_f32; a.stats = stats;
    a.eact = d->epilogue_act;
    return run_fwd(d->dtype, g, a, (hipStream_t)stream);
}

static int conv_wgrad_impl(const pai_conv_desc* d, const void* x1, const void* x2, const void* dy, float* dw,
                           float* dbias, int overwrite, void* stream, const float* pscale = nullptr,
                           const float* pshift = nullptr, int pact = 0);

// ---- prologue: the input read as act(x * scale[c] + shift[c]) -- the BatchNorm + activation of the producing layer applied
// on load, so that its activated tensor is never written (reference models/res_unet.py:143-147: Conv2d -> BatchNorm2d -> ReLU ->
// Conv2d; the second convolution and its weight gradient read the first one's raw output)
static bool prologue_fwd_ok(const pai_conv_desc* d, const GG& g) {
    FwdArgs a;
    memset(&a, 0, sizeof(a));
    a.y1 = (void*)1;
    a.pscale = a.pshift = (const float*)1;
    a.pact = PAI_ACT_RELU;
    return grouped3_ok(d->dtype, g, a) || pwx_ok(d->dtype, g, a);
}

// weight gradient through a prologue: the pointwise tile kernel, or the grouped 3 x 3 kernel with its workspace registered
static bool prologue_wgrad_ok(const pai_conv_desc* d, const GG& g) {
    if (grouped3_wgrad_ok(d->dtype, g, nullptr)) return wgrad_slab_acquire(grouped3_wgrad_part_bytes(g)) != nullptr;
    return wgrad_pro_ok(d->dtype, g);
}

extern "C" int pai_conv_prologue_ok(const pai_conv_desc* d) {
    GG g;
    if (gg_build_fwd(d, &g)) return 0;
    return prologue_fwd_ok(d, g) && prologue_wgrad_ok(d, g);
}

extern "C" int pai_conv_fwd_pro(const pai_conv_desc* d, const void* x1, const void* w_fwd, const float* bias, void* y_raw,
                                float* stats, const float* pre_scale, const float* pre_shift, int pre_act, void* stream) {
    GG g;
    if (gg_build_fwd(d, &g)) return 1;
    PAI_CHECK(x1 && w_fwd && y_raw && pre_scale && pre_shift, "pai_conv_fwd_pro: null pointer");
    PAI_CHECK(pre_act == PAI_ACT_NONE || pre_act == PAI_ACT_RELU, "pai_conv_fwd_pro: pre_act=%d (none or ReLU)", pre_act);
    PAI_CHECK(d->C2 == 0, "pai_conv_fwd_pro: one source tensor only");
    FwdArgs a;
    memset(&a, 0, sizeof(a));
    a.x1 = x1; a.w = w_fwd; a.bias = bias;
    a.y1 = y_raw;
    a.stats = stats;
    a.pscale = pre_scale; a.pshift = pre_shift; a.pact = pre_act;
    return run_fwd(d->dtype, g, a, (hipStream_t)stream);
}

extern "C" int pai_conv_wgrad_pro(const pai_conv_desc* d, const void* x1, const void* dy, float* dw, float* dbias,
                                  int overwrite, const float* pre_scale, const float* pre_shift, int pre_act, void* stream) {
    PAI_CHECK(pre_scale && pre_shift, "pai_conv_wgrad_pro: null prologue");
    PAI_CHECK(pre_act == PAI_ACT_NONE || pre_act == PAI_ACT_RELU, "pai_conv_wgrad_pro: pre_act=%d (none or ReLU)", pre_act);
    PAI_CHECK(d->C2 == 0, "pai_conv_wgrad_pro: one source tensor only");
    return conv_wgrad_impl(d, x1, nullptr, dy, dw, dbias, overwrite ? 1 : 0, stream, pre_scale, pre_shift, pre_act);
}

extern "C" int pai_conv_dgrad(const pai_conv_desc* d, const void* dy, const void* w_dgrad,
                              void* dx1, void* dx2, int only_c2, void* stream) {
    GG g;
    if (gg_build_dgrad(d, &g)) return 1;
    PAI_CHECK(dy && w_dgrad, "pai_conv_dgrad: null dy / w");
    PAI_CHECK(only_c2 ? (dx2 && d->C2 > 0) : (dx1 != nullptr), "pai_conv_dgrad: missing output");
    PAI_CHECK(d->C2 == 0 || dx2, "pai_conv_dgrad: C2 > 0 but dx2 is null");
    FwdArgs a;
    memset(&a, 0, sizeof(a));
    a.x1 = dy; a.w = w_dgrad;
    a.y1 = dx1; a.y2 = dx2;
    a.skip_d1 = only_c2;
    return run_fwd(d->dtype, g, a, (hipStream_t)stream);
}

static bool dgrad_store_fusable(const pai_conv_desc* d, const GG& g, const FwdArgs& a) {
    if (grouped3_ok(d->dtype, g, a)) return false;
    if (pw_ok(d->dtype, g, a)) return true;
    return !thin_fwd_ok(d->dtype, g, a) && !thin_dgrad_ok(d->dtype, g, a) && !fwd_rowdot_ok(g, a) &&
           use_mfma(d->dtype, g, a);
}

extern "C" int pai_conv_dgrad_act(const pai_conv_desc* d, const void* dy, const void* w_dgrad,
                                  void* dx1, void* dx2, const void* a1, int act1, void* stream) {
    GG g;
    if (gg_build_dgrad(d, &g)) return 1;
    PAI_CHECK(dy && w_dgrad && dx1 && a1, "pai_conv_dgrad_act: null pointer");
    PAI_CHECK(d->C2 == 0 || dx2, "pai_conv_dgrad_act: C2 > 0 but dx2 is null");
    PAI_CHECK(act1 == PAI_ACT_LRELU || act1 == PAI_ACT_RELU || act1 == PAI_ACT_NONE, "pai_conv_dgrad_act: act1=%d", act1);
    FwdArgs a;
    memset(&a, 0, sizeof(a));
    a.x1 = dy; a.w = w_dgrad;
    a.y1 = dx1; a.y2 = dx2;
    hipStream_t s = (hipStream_t)stream;
    // (the PatchGAN head's own kernel applies the activation derivative in its store too: nothing else is asked of it here)
    const bool fused = dgrad_store_fusable(d, g, a) || (!thin_fwd_ok(d->dtype, g, a) && head_dgrad_ok(d->dtype, g, a));
    if (fused) { a.bz = a1; a.bact1 = act1; }
    int rc = run_fwd(d->dtype, g, a, s);
    if (rc || fused || act1 == PAI_ACT_NONE) return rc;
    // kernels without the fused store: the same product as a second pass, in place
    return pai_act_bwd(d->dtype, dx1, act1, nullptr, PAI_ACT_NONE, a1, (int64_t)g.N * g.OH * g.OW * g.D1, dx1, stream);
}

extern "C" int pai_conv_dgrad_bn_rows_max(const pai_conv_desc* d) {
    GG g;
    if (gg_build_dgrad(d, &g)) return -1;
    int fused = cdiv(g.M, 16) * g.nphase;
    if (pw_rows(g) > fused) fused = pw_rows(g);
    if (pwx_rows(g) > fused) fused = pwx_rows(g);
    const int twopass = pai_bn_bwd_partial_rows((int64_t)g.N * g.OH * g.OW);
    return fused > twopass ? fused : twopass;
}

int bn_bwd_reduce_affine(int dtype, void* g1_du, int act1, const void* g2, int act2, const void* z, int64_t M, int C,
                         const float* scale, const float* shift, const float* mean, const float* rstd,
                         float* partials, hipStream_t s);

extern "C" int pai_conv_dgrad_bn(const pai_conv_desc* d, const void* dy, const void* w_dgrad, void* dx1,
                                 void* dx2, const pai_bwd_epilogue* e, int* partial_rows, void* stream) {
    GG g;
    if (gg_build_dgrad(d, &g)) return 1;
    PAI_CHECK(dy && w_dgrad && dx1 && e && e->z, "pai_conv_dgrad_bn: null pointer");
    PAI_CHECK(d->C2 == 0 || dx2, "pai_conv_dgrad_bn: C2 > 0 but dx2 is null");
    PAI_CHECK((e->scale == nullptr) == (e->shift == nullptr), "pai_conv_dgrad_bn: scale and shift go together");
    PAI_CHECK(!e->partials || (e->mean && e->rstd && partial_rows), "pai_conv_dgrad_bn: partials need mean, rstd and partial_rows");
    for (int act : {e->act1, e->act2})
        PAI_CHECK(act == PAI_ACT_LRELU || act == PAI_ACT_RELU || act == PAI_ACT_NONE, "pai_conv_dgrad_bn: act=%d", act);
    FwdArgs a;
    memset(&a, 0, sizeof(a));
    a.x1 = dy; a.w = w_dgrad;
    a.y1 = dx1; a.y2 = dx2;
    hipStream_t s = (hipStream_t)stream;
    const int64_t M = (int64_t)g.N * g.OH * g.OW;
    if (dgrad_store_fusable(d, g, a)) {
        a.bz = e->z; a.badd = e->add;
        a.bscale = e->scale; a.bshift = e->shift; a.bmean = e->mean; a.brstd = e->rstd;
        a.bact1 = e->act1; a.bact2 = e->act2;
        a.bpart = e->partials;
        if (partial_rows)
            *partial_rows = !e->partials ? 0 : (pw_ok(d->dtype, g, a) ? pw_rows(g) : pwx_ok(d->dtype, g, a) ? pwx_rows(g)
                                                                                      : fwd_mfma_mtiles(g) * g.nphase);
        return run_fwd(d->dtype, g, a, s);
    }
    if (e->partials) {
        // thin -> wide input gradient (the head of the U-Net): the pass rides on thin_fwd2_k's store
        if (const int rows = thin_fwd_bwd_rows(d->dtype, g, a, e->act1, e->add, e->scale)) {
            a.bz = e->z; a.bmean = e->mean; a.brstd = e->rstd; a.bpart = e->partials; a.bact1 = e->act1;
            *partial_rows = rows;
            return run_fwd(d->dtype, g, a, s);
        }
    }
    // other kernel families: plain input gradient, then the same arithmetic as a second pass in place
    int rc = run_fwd(d->dtype, g, a, s);
    if (rc) return rc;
    if (e->partials) {
        *partial_rows = pai_bn_bwd_partial_rows(M);
        return bn_bwd_reduce_affine(d->dtype, dx1, e->act1, e->add, e->act2, e->z, M, g.D1, e->scale, e->shift,
                                    e->mean, e->rstd, e->partials, s);
    }
    if (partial_rows) *partial_rows = 0;
    PAI_CHECK(e->scale == nullptr, "pai_conv_dgrad_bn: affine pre-activation without partial sums is only built for the matrix-core kernels");
    return pai_act_bwd(d->dtype, dx1, e->act1, e->add, e->act2, e->z, M * g.D1, dx1, stream);
}

// Convolution + BatchNorm2d(train) + activation.  One entry point so that the library may pick the launch sequence; today
// every layer runs pai_conv_fwd(stats) -> pai_bn_finalize -> pai_bn_apply exactly as the three separate calls would.
extern "C" int pai_conv_fwd_bn(const pai_conv_desc* d, const void* x1, const void* x2, const void* w_fwd, const float* bias,
                               void* z, void* a_out, int act, const pai_bn_train* bn, float* stats, void* stream) {
    GG g;
    if (gg_build_fwd(d, &g)) return 1;
    PAI_CHECK(x1 && w_fwd && z && a_out && bn && stats, "pai_conv_fwd_bn: null pointer");
    PAI_CHECK(d->C2 == 0 || x2, "pai_conv_fwd_bn: C2 > 0 but x2 is null");
    PAI_CHECK(bn->mean && bn->rstd && bn->scale && bn->shift, "pai_conv_fwd_bn: mean / rstd / scale / shift outputs are required");
    PAI_CHECK(act == PAI_ACT_NONE || act == PAI_ACT_RELU || act == PAI_ACT_LRELU, "pai_conv_fwd_bn: act=%d", act);
    FwdArgs a;
    memset(&a, 0, sizeof(a));
    a.x1 = x1; a.x2 = x2; a.w = w_fwd; a.bias = bias;
    a.y1 = z; a.stats = stats;
    a.eact = PAI_ACT_NONE;
    hipStream_t s = (hipStream_t)stream;
    const int64_t count = (int64_t)g.N * g.OH * g.OW;
    if (int rc = run_fwd(d->dtype, g, a, s)) return rc;
    const int rows = pai_conv_fwd_stats_rows(d);
    if (bn_fuse_small_ok(rows, count, g.Cout))      // the U-Net bottleneck: finalize + apply as ONE launch, same results
        return launch_bn_fin_apply(d->dtype, stats, rows, g.Cout, count, bn->gamma, bn->beta, bn->eps, bn->momentum,
                                   bn->n_updates, bn->running_mean, bn->running_var, bn->num_batches_tracked, bn->mean,
                                   bn->rstd, bn->scale, bn->shift, z, act, a_out, s);
    if (int rc = pai_bn_finalize(stats, rows, g.Cout, count, bn->gamma, bn->beta, bn->eps, bn->momentum, bn->n_updates,
                                 bn->running_mean, bn->running_var, bn->num_batches_tracked, bn->mean, bn->rstd, bn->scale,
                                 bn->shift, stream)) return rc;
    return pai_bn_apply(d->dtype, z, count, g.Cout, bn->scale, bn->shift, act, a_out, stream);
}

// Always 0 since ABI 131: the single column-owner finish launch of rounds 3-4 (gg_finish.hip) was slower than the three
// launches it replaced and is gone; the entry point stays so that ABI-130 callers link.
extern "C" int pai_conv_bn_fused(const pai_conv_desc* d, int op) {
    GG g;
    if (op == 1 ? gg_build_dgrad(d, &g) : gg_build_fwd(d, &g)) return -1;
    return 0;
}

// Input gradient + the COMPLETE BatchNorm backward of the layer that produced the input: dz (gradient w.r.t. that
// layer's convolution output), sums, dgamma / dbeta:
// pai_conv_dgrad_bn (du into `du_scratch`, partial rows into e->partials) -> pai_bn_bwd_finalize -> pai_bn_bwd_apply.
extern "C" int pai_conv_dgrad_bn_apply(const pai_conv_desc* d, const void* dy, const void* w_dgrad, void* du_scratch,
                                       void* dx2, const pai_bwd_epilogue* e, const float* gamma, float* sums,
                                       float* dgamma, float* dbeta, void* dz, void* stream) {
    GG g;
    if (gg_build_dgrad(d, &g)) return 1;
    PAI_CHECK(dy && w_dgrad && du_scratch && e && e->z && e->mean && e->rstd && e->partials && sums && dz,
              "pai_conv_dgrad_bn_apply: null pointer");
    PAI_CHECK(d->C2 == 0 || dx2, "pai_conv_dgrad_bn_apply: C2 > 0 but dx2 is null");
    PAI_CHECK((e->scale == nullptr) == (e->shift == nullptr), "pai_conv_dgrad_bn_apply: scale and shift go together");
    FwdArgs a;
    memset(&a, 0, sizeof(a));
    a.x1 = dy; a.w = w_dgrad;
    a.y1 = du_scratch; a.y2 = dx2;
    hipStream_t s = (hipStream_t)stream;
    int rows = 0;
    if (int rc = pai_conv_dgrad_bn(d, dy, w_dgrad, du_scratch, dx2, e, &rows, stream)) return rc;
    const int64_t Mo = (int64_t)g.N * g.OH * g.OW;
    if (bn_fuse_small_ok(rows, Mo, g.D1))
        return launch_bn_bwd_fin_apply(d->dtype, e->partials, rows, g.D1, sums, dgamma, dbeta, du_scratch, e->z, Mo, e->mean,
                                       e->rstd, gamma, dz, s);
    if (int rc = pai_bn_bwd_finalize(e->partials, rows, g.D1, sums, dgamma, dbeta, stream)) return rc;
    return pai_bn_bwd_apply(d->dtype, du_scratch, e->z, Mo, g.D1, e->mean, e->rstd, gamma, sums, dz, stream);
}


extern "C" int pai_conv_wgrad(const pai_conv_desc* d, const void* x1, const void* x2,
                              const void* dy, float* dw, float* dbias, void* stream) {
    return conv_wgrad_impl(d, x1, x2, dy, dw, dbias, 0, stream);
}

extern "C" int pai_conv_wgrad_overwrite(const pai_conv_desc* d, const void* x1, const void* x2,
                                        const void* dy, float* dw, float* dbias, void* stream) {
    return conv_wgrad_impl(d, x1, x2, dy, dw, dbias, 1, stream);
}

// dw = ..., dbias += ...: the weights need no caller zeroing, the bias gradient is ADDED to what the caller cleared (one
// multi-tensor clear of all small gradient segments per backward pass instead of one fill launch per layer)
extern "C" int pai_conv_wgrad_overwrite_w(const pai_conv_desc* d, const void* x1, const void* x2,
                                          const void* dy, float* dw, float* dbias, void* stream) {
    return conv_wgrad_impl(d, x1, x2, dy, dw, dbias, 2, stream);
}

static int conv_wgrad_impl(const pai_conv_desc* d, const void* x1, const void* x2, const void* dy, float* dw,
                           float* dbias, int overwrite, void* stream, const float* pscale, const float* pshift, int pact) {
    GG g;
    if (gg_build_fwd(d, &g)) return 1;
    PAI_CHECK(x1 && dy && dw, "pai_conv_wgrad: null pointer");
    PAI_CHECK(d->C2 == 0 || x2, "pai_conv_wgrad: C2 > 0 but x2 is null");
    WgradArgs a;
    a.x1 = x1; a.x2 = x2; a.dy = dy; a.dw = dw; a.dbias = dbias; a.overwrite = 0; a.overwrite_bias = 0; a.slab = nullptr;
    a.pscale = pscale; a.pshift = pshift; a.pact = pact;
    hipStream_t s = (hipStream_t)stream;
    if (pscale && grouped3_wgrad_ok(d->dtype, g, dbias)) {
        float* part = wgrad_slab_acquire(grouped3_wgrad_part_bytes(g));
        PAI_CHECK(part, "pai_conv_wgrad_pro: the grouped 3 x 3 weight gradient needs its workspace (pai_set_wgrad_workspace)");
        a.overwrite = overwrite != 0;
        return launch_grouped3_wgrad(g, a, part, s);
    }
    if (pscale) {
        PAI_CHECK(wgrad_pro_ok(d->dtype, g), "pai_conv_wgrad_pro: this layer takes no prologue (ask pai_conv_prologue_ok)");
        if (overwrite && wgrad_mfma_can_overwrite(g)) {
            a.overwrite = 1;
            a.overwrite_bias = overwrite == 1;
        } else if (overwrite) {
            hipError_t e = pai::memset_async(dw, 0, (size_t)g.Cout * g.wtaps * g.Cin * sizeof(float), s);
            PAI_CHECK(e == hipSuccess, "pai_conv_wgrad_pro: hipMemsetAsync: %s", hipGetErrorString(e));
            if (dbias && overwrite == 1) {
                e = pai::memset_async(dbias, 0, (size_t)g.Cout * sizeof(float), s);
                PAI_CHECK(e == hipSuccess, "pai_conv_wgrad_pro: hipMemsetAsync: %s", hipGetErrorString(e));
            }
        }
        return launch_wgrad_mfma(g, a, s);
    }
    if (grouped3_wgrad_ok(d->dtype, g, dbias)) {      // block-diagonal 3 x 3 filter: the diagonal blocks only (gg_group.hip)
        float* part = wgrad_slab_acquire(grouped3_wgrad_part_bytes(g));
        if (part) {
            a.overwrite = overwrite != 0;
            return launch_grouped3_wgrad(g, a, part, s);
        }
    }
    if (overwrite) {
        const bool thin = thin_wgrad_conv_ok(d->dtype, g) || thin_wgrad_convt_ok(d->dtype, g) ||
                          thin_wgrad_conv1_ok(d->dtype, g) || thin_wgrad_conv3_ok(d->dtype, g) ||
                          thin_wgrad_conv3t_ok(d->dtype, g);
        const bool rowdot = g.Cout <= 2 && (g.ntaps == 4 || g.ntaps == 9 || g.ntaps == 16) && (g.C1 % 8) == 0 && (g.C2 % 8) == 0;
        if (!thin && !rowdot && wgrad_mfma_ok(d->dtype, g) && wgrad_mfma_can_overwrite(g)) {
            a.overwrite = 1;       // every element has exactly one writer: plain stores, nothing to clear
            a.overwrite_bias = overwrite == 1;
        } else {                   // the accumulating kernels: clear first
            hipError_t e = pai::memset_async(dw, 0, (size_t)g.Cout * g.wtaps * g.Cin * sizeof(float), s);
            PAI_CHECK(e == hipSuccess, "pai_conv_wgrad_overwrite: hipMemsetAsync: %s", hipGetErrorString(e));
            if (dbias && overwrite == 1) {
                e = pai::memset_async(dbias, 0, (size_t)g.Cout * sizeof(float), s);
                PAI_CHECK(e == hipSuccess, "pai_conv_wgrad_overwrite: hipMemsetAsync: %s", hipGetErrorString(e));
            }
        }
    }
    if (thin_wgrad_conv_ok(d->dtype, g)) return launch_thin_wgrad_conv(g, a, s);
    if (thin_wgrad_convt_ok(d->dtype, g)) return launch_thin_wgrad_convt(g, a, s);
    if (thin_wgrad_conv1_ok(d->dtype, g)) return launch_thin_wgrad_conv1(g, a, s);
    if (thin_wgrad_conv3_ok(d->dtype, g)) return launch_thin_wgrad_conv3(g, a, s);
    if (thin_wgrad_conv3t_ok(d->dtype, g)) return launch_thin_wgrad_conv3t(g, a, s);
    if (g.Cout <= 2 && (g.ntaps == 4 || g.ntaps == 9 || g.ntaps == 16) && (g.C1 % 8) == 0 && (g.C2 % 8) == 0) {
        int chunks = g.Cin / 8;
        if (chunks <= 256 && (chunks & (chunks - 1)) == 0) return launch_wgrad_rowdot(d->dtype, g, a, s);
    }
    if (wgrad_mfma_ok(d->dtype, g)) return launch_wgrad_mfma(g, a, s);
    return launch_wgrad_simt(d->dtype, g, a, s);
}

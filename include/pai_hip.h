/*
 * pai_hip.h -- C ABI of libpai_hip.so, the MI355X (gfx950) implementation of the
 * Pix2Pix / U-Net training hot path of cristianpjensen/thesis-pai-reconstruction.
 *
 * The reference has no FFI of its own: its hot path sits behind torch.nn modules
 * that dispatch to ATen.  Each entry point below replaces one ATen operator family
 * at the call sites cited next to it (paths relative to the reference repo).
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error; pai_last_error()
 *     returns a thread-local message.  Nothing throws across the ABI.
 *   - all pointers are DEVICE pointers unless a parameter says "host".
 *   - nothing is allocated or freed on behalf of the caller: outputs and
 *     workspaces are caller-provided (sizes from the *_rows / *_bytes queries).
 *   - no process-wide mutable state except the per-device HANDLE (pai_create): it owns the
 *     caller-provided split-K workspace and scratch; launches use the active handle of the
 *     calling thread's current HIP device.  One host thread drives a device at a time.
 *   - every launch goes on the caller's hipStream_t (passed as void*), is
 *     asynchronous and never synchronises: all entry points are hipGraph-capturable.
 *   - activations are NHWC ("channels last"): [N][H][W][C], C contiguous.
 *   - packed filter layout ("fwd pack"):  [Cout][k*k][Cin]  (tap = kh*k+kw)
 *     transposed layout   ("dgrad pack"): [Cin][k*k][Cout]
 *     for BOTH Conv2d and ConvTranspose2d, where Cin/Cout are the layer's
 *     input/output channels.  Weight gradients are produced in the fwd pack, fp32.
 *   - dtype selects the STORAGE type of activations and packed filters
 *     (PAI_F32 or PAI_BF16); accumulation, statistics and gradients of
 *     parameters are always fp32.
 */
#ifndef PAI_HIP_H
#define PAI_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PAI_F32 0
#define PAI_BF16 1

#define PAI_ACT_NONE 0
#define PAI_ACT_LRELU 1   /* LeakyReLU(0.2)  models/pix2pix.py:62, models/wrapper.py:205 */
#define PAI_ACT_RELU 2    /* ReLU            models/pix2pix.py:98 */
#define PAI_ACT_TANH 3    /* Tanh            models/pix2pix.py:196 */

const char* pai_last_error(void);
/* 100: round 1.  110: per-device handles, pai_set_tunable, pai_adam_dev, pai_scalar_take / pai_metrics_take,
 * pai_pack_weights_multi; pai_bn_bwd_reduce accepts du = NULL.  120: weight-gradient workspace
 * (pai_set_wgrad_workspace), pai_build_flags, PAI_TUNABLE_UNSET; pai_conv_desc.pack_flags bits other than 0-1 and
 * .reserved (ABI 130: .hints) are CHECKED to be zero (descriptors must be zero-initialised; a 100 caller that did so runs unchanged),
 * pai_conv_fwd_bn / pai_conv_dgrad_bn_apply / pai_conv_bn_fused, pai_conv_wgrad_overwrite_w, pai_adam_multi_dev.
 * 121: pai_adam_pack, pai_bn_bwd_apply_affine (pai_bn_bwd_reduce_affine accepts du = NULL); with groups > 1 the weight
 * gradient of a 3 x 3 layer defines the diagonal 16-channel blocks of dw only.  130: launch plans (pai_plan_*,
 * pai_stream_wait, pai_event_*), pai_zero_multi, pai_scale; pai_pack_frag and the pack_flags bits are gone (removed
 * experiment kernels: pack_flags MUST be zero); pai_conv_desc.reserved became .hints (PAI_HINT_SOLO).
 * 131: pai_lerp_multi (the EMA update of callbacks/ema.py), PAI_TUNE_<name> environment defaults of the tunables.
 * 132: input prologue (pai_conv_prologue_ok, pai_conv_fwd_pro, pai_conv_wgrad_pro); pai_instnorm_fwd / _bwd; pai_bn2_bwd_reduce / _apply; pai_bn_stats_buffer_rows grows for
 * layers with more than 2048 partial rows (callers that size the buffer through it need no change). */
int pai_version(void);
/* Build-option bits.  0 since ABI 130: bit 0 used to announce the round-2 experiment kernels (and pai_pack_frag), which
 * were removed from the library. */
int pai_build_flags(void);
/* Device properties of the current HIP device (host out-params). */
int pai_device_info(int* cu_count, int* lds_bytes, char* arch_name, int arch_name_len);

/* Kernel-selection switches (host side, process-wide), e.g. "wgrad3" = 0 routes the dense weight gradients back to
 * gg_wgrad_patch_k, "wgrad_slab" = 0 makes their pixel splits meet through fp32 atomics again.  Every switch has a
 * built-in default (the experiment kernels default to OFF); value PAI_TUNABLE_UNSET returns a switch to it.  For
 * tests that pin the kernel a call runs and for A/B timing in one process; production code never needs it. */
#define PAI_TUNABLE_UNSET (-2147483647 - 1)
int pai_set_tunable(const char* name, int value);

/* ---------------------------------------------------------------------------
 * Convolution family.  One descriptor serves Conv2d and ConvTranspose2d.
 * Replaces aten::convolution / aten::convolution_backward issued by
 *   nn.Conv2d(k4,s2,p1)            models/pix2pix.py:63-69,141-147  models/wrapper.py:197-203
 *   nn.Conv2d(512,1,k4,s1,p1)      models/wrapper.py:233
 *   nn.ConvTranspose2d(k4,s2,p1)   models/pix2pix.py:99-105,186-192
 * and the torch.cat in front of them (models/pix2pix.py:212, models/wrapper.py:237):
 * the input may be given as two tensors x1|x2 that are read as if concatenated
 * along C (x1's channels first), and an input gradient may be written to two
 * tensors the same way.
 * ------------------------------------------------------------------------- */
typedef struct pai_conv_desc {
    int32_t dtype;        /* PAI_F32 | PAI_BF16 */
    int32_t transposed;   /* 0 = Conv2d, 1 = ConvTranspose2d */
    int32_t N, H, W;      /* input batch / height / width */
    int32_t C1, C2;       /* input channels taken from x1 and x2 (C2 = 0: x2 unused) */
    int32_t Cout;         /* output channels */
    int32_t kernel;       /* 4; 1: pointwise Conv2d of the attention gates (models/attention_unet.py:72-84;
                             stride 1, pad 0, one weight tap); 3: "same" Conv2d of the residual U-Net
                             (models/res_unet.py:59; stride 1, pad 1, 9 taps) */
    int32_t stride;       /* 2 (or 1 for Conv2d) */
    int32_t pad;          /* 1 (0 with kernel 1) */
    int32_t relu1, relu2; /* apply ReLU to x1 / x2 while loading (fused nn.ReLU of the
                             decoder block, models/pix2pix.py:98) */
    int32_t epilogue_act; /* PAI_ACT_* applied to y_act / y_f32 */
    int32_t groups;       /* 0 or 1: dense.  > 1 (kernel = 3, Conv2d, C1 = Cout, C2 = 0): the filter packs are the
                             BLOCK-DIAGONAL dense form of a grouped convolution (nn.Conv2d(groups=32) of
                             ResidualBlockNeXt, models/res_unet.py:151-157) whose groups do not straddle 16-channel
                             slices; forward / input gradient may then skip the zero blocks, and the weight gradient
                             is only defined on the 16-channel diagonal blocks that contain the groups (with the
                             weight-gradient workspace registered nothing else of dw is written, not even by the
                             _overwrite forms; without it the dense kernel leaves cross-group products there).
                             Otherwise a hint: every kernel family computes the same result from the dense packs. */
    int32_t pack_flags;   /* MUST be zero (checked).  ABI 110-121 announced fragment-major pack copies here for an
                             experiment kernel that was removed in ABI 130. */
    int32_t hints;        /* PAI_HINT_* bits; the others MUST be zero (checked).  Hints change launch geometry only, never
                             results beyond summation order (and not even that where the library sums in a fixed order). */
} pai_conv_desc;
/* The launch will run ALONE on the device (no other stream has work beside it): the weight gradient of the last layers
 * of a backward pass, when the input-gradient chain has already ended.  The library then sizes the grid for the whole
 * chip (two workgroups per CU) instead of for co-scheduling with the input-gradient stream (one per CU). */
#define PAI_HINT_SOLO 1

/* Output spatial size of the layer. */
int pai_conv_out_hw(const pai_conv_desc* d, int* OH, int* OW);
/* Number of partial-statistics rows pai_conv_fwd writes when stats != NULL, and the
 * number of rows the caller must allocate for that buffer (the tail is scratch for
 * pai_bn_finalize's two-stage fp64 reduction). */
int pai_conv_fwd_stats_rows(const pai_conv_desc* d);
/* Upper bound of the above over every launch configuration (it depends on whether the split-K
 * scratch is registered): allocate pai_bn_stats_buffer_rows(pai_conv_fwd_stats_rows_max(d)) rows. */
int pai_conv_fwd_stats_rows_max(const pai_conv_desc* d);
int pai_bn_stats_buffer_rows(int rows);

/* Kernel family a call with this descriptor runs (for profiling / roofline accounting):
 * op 0 = forward, 1 = input gradient, 2 = weight gradient.
 * returns 0 vector-ALU tile kernel, 1 row-dot kernel, 2 bf16 MFMA 128-wide tile, 3 bf16 MFMA
 * 64-wide tile, 4 thin-layer MFMA kernels, 5 small-channel (16 / 32) MFMA kernel, < 0 on error. */
int pai_conv_kernel_id(const pai_conv_desc* d, int op);
/* Symbol (as rocprofv3 --kernel-trace prints it, without "void " and the argument list) of the main
 * kernel such a call launches, e.g. "gg_fwd_patch_k<256, 128, true>"; bench.py keys its per-kernel
 * roofline on it so that the figure can be checked against profiles/ *_kernel_stats.csv.
 * Families without a single dominant kernel report their family name.  Returns 0, < 0 on error. */
int pai_conv_kernel_name(const pai_conv_desc* d, int op, char* name, int name_len);

/* ---------------------------------------------------------------------------
 * Per-device handle.  Owns the two caller-provided work buffers below; nothing else in the library is
 * stateful.  pai_create(device_id) returns a handle for that HIP device (the first one created for a
 * device becomes its ACTIVE handle); pai_bind(handle) makes a handle the active one of its device;
 * every launch uses the active handle of the calling thread's current device; pai_destroy releases the
 * handle (never the buffers).  A device without a handle runs every layer un-split / on the fallback
 * kernels.
 * STREAM CONTRACT of the two buffers (per handle):
 *   workspace  used by forward / input-gradient calls whose GEMM is split over K: those calls must be
 *              ordered with respect to each other (one stream, or events).
 *   scratch    forward / input-gradient calls of the thin layers use its HEAD, weight-gradient calls its
 *              TAIL: calls within each class must be ordered, the two classes may run concurrently
 *              (the engine runs weight gradients on a second stream).
 *   wgrad workspace  weight-gradient calls of the dense layers: ordered with respect to each other.
 * ------------------------------------------------------------------------- */
int pai_create(int device_id, void** handle_out);
int pai_bind(void* handle);
int pai_destroy(void* handle);
int pai_handle_set_workspace(void* handle, void* zeroed_device_memory, int64_t bytes);
int pai_handle_set_scratch(void* handle, void* device_memory, int64_t bytes);
int pai_handle_set_wgrad_workspace(void* handle, void* device_memory, int64_t bytes);

/* Split-K workspace.  Layers whose GEMM has few output tiles but a long reduction (the U-Net
 * bottleneck: M <= 1024 rows, K up to 8192) are split over K; every split writes its fp32 partial
 * tile into its own slab of the handle's workspace (plain stores, no atomics) and a finish kernel sums
 * the slabs in a fixed order.  Size: at least max(pai_conv_workspace_bytes(desc, op)) over the forward
 * (op 0) and input-gradient (op 1) calls that will be made; contents need not be preserved between
 * calls.  Calls that would need more than is registered run un-split.
 * pai_set_workspace / pai_set_scratch: the same on the active handle of the current device, which is
 * created on first use (convenience for callers with one model per device). */
int pai_set_workspace(void* zeroed_device_memory, int64_t bytes);
int64_t pai_conv_workspace_bytes(const pai_conv_desc* d, int op);
/* General (dirty) scratch of the handle: the wide->thin layers (ConvTranspose2d(128,1) head, input gradient of
 * the first discriminator conv) run as a skinny GEMM into fp32 scratch followed by a col2im pass.
 * Their weight gradients (op 2) collect one partial tile per workgroup there and add them in a second
 * pass.  Forward / input-gradient calls use the HEAD of the buffer, weight-gradient calls its TAIL, so
 * that the two may run on different streams: register at least
 *   max over op 0,1 of pai_conv_scratch_bytes(desc, op)  +  max of pai_conv_scratch_bytes(desc, 2).
 * Without it those layers fall back to the slower row-dot kernel / to fp32 atomics. */
int pai_set_scratch(void* device_memory, int64_t bytes);
int64_t pai_conv_scratch_bytes(const pai_conv_desc* d, int op);
/* Weight-gradient workspace (dirty, fp32).  The dense k4 s2 weight gradients are split over the pixels so that the
 * chip is full; every split stores its partial dW into its own slab of this buffer (plain stores) and a second kernel
 * adds the slabs in split order -- deterministic, and without the ~1.3 TB/s memory-side float atomics that the splits
 * otherwise meet through.  Register at least max(pai_conv_wgrad_workspace_bytes(desc)) over the layers; a call that
 * would need more than is registered falls back to atomics.  Stream contract: weight-gradient calls that use it must
 * be ordered with respect to each other (the engine issues them on one side stream). */
int pai_set_wgrad_workspace(void* device_memory, int64_t bytes);
int64_t pai_conv_wgrad_workspace_bytes(const pai_conv_desc* d);

/* y = conv(act(x1|x2), w) + bias.
 *   w_fwd   : fwd pack, storage dtype
 *   bias    : fp32 [Cout] or NULL
 *   y_raw   : storage dtype [N][OH][OW][Cout], pre-activation value, or NULL
 *   y_act   : storage dtype, epilogue_act(y), or NULL
 *   y_f32   : fp32, epilogue_act(y), or NULL
 *   stats   : fp32 [rows][2][Cout] per-tile partial (sum, sum of squares) of the
 *             pre-activation fp32 value for BatchNorm, or NULL */
int pai_conv_fwd(const pai_conv_desc* d, const void* x1, const void* x2, const void* w_fwd,
                 const float* bias, void* y_raw, void* y_act, float* y_f32, float* stats,
                 void* stream);

/* dx1|dx2 = conv_backward_input(dy, w).  w_dgrad: dgrad pack, storage dtype.
 * dy: storage dtype [N][OH][OW][Cout].  dx2 may be NULL when C2 == 0.
 * only_c2: if non-zero only dx2 is produced (gradient w.r.t. the second input
 * only -- the generator image in Discriminator.forward, models/wrapper.py:237). */
int pai_conv_dgrad(const pai_conv_desc* d, const void* dy, const void* w_dgrad, void* dx1,
                   void* dx2, int only_c2, void* stream);

/* pai_conv_dgrad followed by the activation backward of the layer that produced x1, in one pass:
 *   dx1 = act1'(a1) * conv_backward_input(dy, w)[:, :C1],  dx2 as pai_conv_dgrad
 * a1: storage dtype, shaped like dx1 (the stored activation whose sign carries the derivative:
 * nn.LeakyReLU(0.2) / nn.ReLU in front of the next block, models/wrapper.py:205,
 * models/pix2pix.py:62,98).  Bit-identical to pai_conv_dgrad + pai_act_bwd(dx1, act1, a1); the
 * matrix-core kernels apply it in their store, the others run the second pass themselves. */
int pai_conv_dgrad_act(const pai_conv_desc* d, const void* dy, const void* w_dgrad, void* dx1,
                       void* dx2, const void* a1, int act1, void* stream);

/* pai_conv_dgrad with the backward of the layer that PRODUCED x1 fused into the store of dx1 --
 * the activation in front of this convolution and, optionally, the first pass of the BatchNorm
 * backward of that layer (nn.BatchNorm2d of EncoderBlock / DecoderBlock, models/pix2pix.py:70,106;
 * nn.LeakyReLU / nn.ReLU at :62,98; replaces the threshold_backward + native_batch_norm_backward
 * reductions autograd would run on the materialised gradient):
 *   pre  = z * scale + shift            (scale == NULL: pre = z, e.g. a stored activation)
 *   dx1  = act1'(pre) * conv_backward_input(dy, w)[:, :C1] + act2'(pre) * add
 *   partials[row] = (sum dx1, sum dx1 * (z - mean) * rstd) per channel over the rows of one output
 *                   tile, from the value as stored; *partial_rows rows are written.
 * dx2 as pai_conv_dgrad.  Continue with pai_bn_bwd_finalize(partials, *partial_rows, C1, ...) and
 * pai_bn_bwd_apply(dx1, z, ...).  The matrix-core kernels do this in their store (no extra pass
 * over dx1); the other kernel families run the same arithmetic as a second pass in place.
 * z, add: storage dtype, shaped like dx1.  scale/shift/mean/rstd: fp32 [C1]. */
typedef struct pai_bwd_epilogue {
    const void* z;
    const void* add;       /* or NULL */
    const float* scale;    /* or NULL (then shift is NULL too) */
    const float* shift;
    const float* mean;     /* needed with partials */
    const float* rstd;
    float* partials;       /* fp32 [pai_conv_dgrad_bn_rows_max(d)][2][C1], or NULL: no sums */
    int32_t act1, act2;    /* PAI_ACT_NONE | PAI_ACT_RELU | PAI_ACT_LRELU */
} pai_bwd_epilogue;
int pai_conv_dgrad_bn_rows_max(const pai_conv_desc* d);
int pai_conv_dgrad_bn(const pai_conv_desc* d, const void* dy, const void* w_dgrad, void* dx1,
                      void* dx2, const pai_bwd_epilogue* e, int* partial_rows, void* stream);

/* Convolution + BatchNorm2d in training mode + activation (Conv2d / ConvTranspose2d -> nn.BatchNorm2d -> ReLU /
 * LeakyReLU, reference models/pix2pix.py:63-70,99-106) as ONE call: z = conv + bias (storage dtype, kept for the
 * backward pass), batch statistics of z, running statistics advanced n_updates times, a = act(z * scale + shift).
 * Same results as pai_conv_fwd(stats) -> pai_bn_finalize -> pai_bn_apply, which is what runs (ABI 120-130 ended the
 * U-Net bottleneck's split-K layers in one column-owner finish launch instead; measured slower, removed in 131).
 * `stats`: the buffer pai_conv_fwd would take. */
typedef struct pai_bn_train {
    const float* gamma;            /* or NULL (= 1) */
    const float* beta;             /* or NULL (= 0) */
    float eps, momentum;
    int32_t n_updates;             /* how many times the running statistics advance (Q6: twice per GAN step) */
    float* running_mean;           /* or NULL */
    float* running_var;
    int64_t* num_batches_tracked;  /* or NULL */
    float *mean, *rstd, *scale, *shift;   /* fp32 [Cout] outputs */
} pai_bn_train;
int pai_conv_fwd_bn(const pai_conv_desc* d, const void* x1, const void* x2, const void* w_fwd, const float* bias,
                    void* z, void* a, int act, const pai_bn_train* bn, float* stats, void* stream);
/* Input gradient + the whole BatchNorm backward of the producer layer: like pai_conv_dgrad_bn, but carried through to
 * dz = gamma * rstd * (du - sum(du) / M - xhat * sum(du xhat) / M), with sums [2][C1], dgamma += , dbeta += written on
 * the way.  `du_scratch` (shaped like dx1) and e->partials are workspace of the call. */
int pai_conv_dgrad_bn_apply(const pai_conv_desc* d, const void* dy, const void* w_dgrad, void* du_scratch, void* dx2,
                            const pai_bwd_epilogue* e, const float* gamma, float* sums, float* dgamma, float* dbeta,
                            void* dz, void* stream);
/* Always 0 since ABI 131 (-1: bad descriptor); ABI 120-130: 1 when the two calls above ended in the single fused finish
 * launch for this layer (op 0 forward, 1 input gradient). */
int pai_conv_bn_fused(const pai_conv_desc* d, int op);

/* dw += conv_backward_weight(act(x1|x2), dy)   (fp32, fwd pack; caller zeroes it first)
 * dbias += sum over N,OH,OW of dy              (fp32 [Cout], or NULL) */
int pai_conv_wgrad(const pai_conv_desc* d, const void* x1, const void* x2, const void* dy,
                   float* dw, float* dbias, void* stream);
/* dw = ..., dbias = ...  (no caller zeroing).  Launches whose every dW element has one writer (the un-split
 * matrix-core kernel: the skinny nn.Linear layers of the TransUNet bottleneck, 1 G fp32 elements per step) store
 * plainly -- no zero-fill pass, no read of dW; the others clear the buffers themselves and accumulate. */
int pai_conv_wgrad_overwrite(const pai_conv_desc* d, const void* x1, const void* x2, const void* dy,
                             float* dw, float* dbias, void* stream);
/* dw = ..., dbias += ...: as above for the weights; the bias gradient is added to what the caller cleared. */
int pai_conv_wgrad_overwrite_w(const pai_conv_desc* d, const void* x1, const void* x2, const void* dy,
                               float* dw, float* dbias, void* stream);

/* ---- input prologue: Conv2d -> BatchNorm2d -> ReLU -> Conv2d without the tensor in the middle
 * (reference models/res_unet.py:143-147, the ResNeXt block).  The SECOND convolution and its weight gradient read the raw
 * output z of the first one as act(z * scale[c] + shift[c]) (scale / shift of pai_bn_finalize or pai_bn_eval_coeffs, act =
 * PAI_ACT_NONE | PAI_ACT_RELU) on load: same roundings as pai_bn_apply followed by the plain call (fma, max, storage-type
 * rounding), one tensor write and two tensor reads less per layer.  bf16, one source tensor, raw output only.
 * pai_conv_prologue_ok: 1 when BOTH calls below serve this layer (today: pointwise layers of 64-256 channels with at least
 * 16384 pixels and power-of-two image sizes); otherwise apply the BatchNorm as a pass of its own. */
int pai_conv_prologue_ok(const pai_conv_desc* d);
int pai_conv_fwd_pro(const pai_conv_desc* d, const void* x1, const void* w_fwd, const float* bias, void* y_raw,
                     float* stats, const float* pre_scale, const float* pre_shift, int pre_act, void* stream);
/* overwrite: 0 = accumulate into dw / dbias, 1 = dw = ..., dbias = ... (as pai_conv_wgrad_overwrite) */
int pai_conv_wgrad_pro(const pai_conv_desc* d, const void* x1, const void* dy, float* dw, float* dbias, int overwrite,
                       const float* pre_scale, const float* pre_shift, int pre_act, void* stream);

/* fp32 master weights (fwd pack) -> storage-dtype fwd pack and/or dgrad pack. */
int pai_pack_weights(int dtype, const float* w_master, int Cout, int taps, int Cin,
                     void* w_fwd_or_null, void* w_dgrad_or_null, void* stream);
/* The same for n layers in one launch (bf16 packs, Cin and Cout multiples of 64): after an optimizer step every pack of
 * a network is stale at once.  All arrays are HOST arrays of n entries; w_fwd[i] or w_dgrad[i] may be NULL. */
int pai_pack_weights_multi(int n, const float* const* w_master, const int32_t* cout, const int32_t* taps,
                           const int32_t* cin, void* const* w_fwd, void* const* w_dgrad, void* stream);

/* ---------------------------------------------------------------------------
 * BatchNorm2d (training and eval).  Replaces aten::native_batch_norm(_backward)
 * issued by nn.BatchNorm2d at models/pix2pix.py:70,106.
 * ------------------------------------------------------------------------- */
/* Reduce the partial statistics of pai_conv_fwd to mean / rstd (biased variance,
 * eps) and fold the affine: scale = gamma*rstd, shift = beta - mean*scale.
 * Running statistics are updated n_updates times with `momentum` and the
 * unbiased variance (the reference runs the generator twice per GAN step:
 * models/wrapper.py:126,147), num_batches_tracked += n_updates. */
int pai_bn_finalize(const float* stats, int rows, int C, int64_t count, const float* gamma,
                    const float* beta, float eps, float momentum, int n_updates,
                    float* running_mean, float* running_var, int64_t* num_batches_tracked,
                    float* mean, float* rstd, float* scale, float* shift, void* stream);
/* Eval mode: scale/shift from the running statistics. */
int pai_bn_eval_coeffs(int C, const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, float eps, float* scale, float* shift,
                       void* stream);
/* out = act(z*scale + shift), elementwise over [M][C]. */
int pai_bn_apply(int dtype, const void* z, int64_t M, int C, const float* scale,
                 const float* shift, int act, void* out, void* stream);
/* out = act(act_a(za*scale_a + shift_a) + (zb*scale_b + shift_b)) over [M][C]: BatchNorm (+ its own ReLU: the ResNeXt
 * block) of the residual branch + BatchNorm of the skip branch (scale_b = shift_b = NULL: identity skip, zb added as it is)
 * + sum + activation, the tail of the residual blocks (reference models/res_unet.py:74,105,160-171,
 * models/trans_unet.py:227-236), in one pass. */
int pai_bn2_add_act(int dtype, const void* za, const float* scale_a, const float* shift_a, const void* zb,
                    const float* scale_b, const float* shift_b, int64_t M, int C, int act_a, int act, void* out,
                    void* stream);
/* Backward, pass 1:  du = act1'(a)*g1 + act2'(a)*g2   (g2 may be NULL)
 *   a   : the stored activated output (its sign gives act'), or NULL when act1 = act2 = none
 *   sums[0][C] = sum(du), sums[1][C] = sum(du * xhat)   with xhat = (z-mean)*rstd
 *   dbeta += sums[0], dgamma += sums[1]   (fp32 [C], either may be NULL)
 * du (storage dtype) is written for pass 2; du may be NULL when g2 = a = NULL and act1 = none (du IS g1 then: hand
 * g1 to pass 2).  `partials` is fp32 workspace of pai_bn_bwd_partial_rows(M) * 2 * C floats. */
int pai_bn_bwd_partial_rows(int64_t M);
int pai_bn_bwd_reduce(int dtype, const void* g1, int act1, const void* g2, int act2, const void* a,
                      const void* z, int64_t M, int C, const float* mean, const float* rstd,
                      void* du, float* partials, float* sums, float* dgamma, float* dbeta,
                      void* stream);
/* pai_bn_bwd_reduce with the activation's sign taken from pre = z*scale + shift (the coefficients pai_bn_finalize
 * produced) instead of from the stored activated output `a`: one tensor read less.  du may be NULL: nothing is stored
 * and pass 2 is pai_bn_bwd_apply_affine, which rebuilds du (one tensor write and nothing else less). */
int pai_bn_bwd_reduce_affine(int dtype, const void* g1, int act1, const void* g2, int act2, const void* z,
                             int64_t M, int C, const float* scale, const float* shift, const float* mean,
                             const float* rstd, void* du, float* partials, float* sums, float* dgamma,
                             float* dbeta, void* stream);
/* Second half of pai_bn_bwd_reduce on its own: reduces `rows` partial rows [2][C] (from
 * pai_conv_dgrad_bn) in fp64 into sums [2][C] and accumulates dbeta += sums[0], dgamma += sums[1]. */
int pai_bn_bwd_finalize(const float* partials, int rows, int C, float* sums, float* dgamma,
                        float* dbeta, void* stream);
/* Backward, pass 2:  dz = gamma*rstd * (du - sums[0]/M - xhat*sums[1]/M). */
int pai_bn_bwd_apply(int dtype, const void* du, const void* z, int64_t M, int C,
                     const float* mean, const float* rstd, const float* gamma,
                     const float* sums, void* dz, void* stream);
/* Pass 2 behind pai_bn_bwd_reduce_affine(..., du = NULL, ...): du = act1'(z*scale + shift) * g1 is rebuilt (and rounded
 * to the storage type, as pass 1 did for its sums), then dz as pai_bn_bwd_apply. */
int pai_bn_bwd_apply_affine(int dtype, const void* g1, int act1, const void* z, int64_t M, int C, const float* scale,
                            const float* shift, const float* mean, const float* rstd, const float* gamma,
                            const float* sums, void* dz, void* stream);
/* du = act1'(a)*g1 + act2'(a)*g2 without BatchNorm (last encoder, discriminator blocks,
 * encoder 0); g2 may be NULL. */
int pai_act_bwd(int dtype, const void* g1, int act1, const void* g2, int act2, const void* a,
                int64_t numel, void* du, void* stream);

/* nn.Dropout2d of the three widest DecoderBlocks (models/pix2pix.py:108,176-179), forward and
 * backward: out[n][p][c] = x[n][p][c] * mask[n][c] on an NHWC tensor [N][HW][C] (may run in place).
 * mask: fp32 [N][C] holding 0 or 1 / (1 - p); drawing it (a Bernoulli sample per sample and channel)
 * is the caller's RNG plumbing. */
int pai_dropout2d(int dtype, const void* x, const float* mask, int N, int64_t HW, int C, void* out,
                  void* stream);

/* ---------------------------------------------------------------------------
 * Residual U-Net building blocks (models/res_unet.py) besides its convolutions (pai_conv_* with kernel 3 / 1)
 * and BatchNorms: nn.MaxPool2d(2) (:199), nearest nn.Upsample(scale_factor=2) (:231), the residual sum with
 * the optional ReLU behind it (:71-74).  NHWC tensors in the storage dtype, C a multiple of 8.
 *   pai_maxpool2      out [N][H/2][W/2][C]; idx (one byte per output element, or NULL) keeps the arg-max
 *                     (first maximum in row-major window order, as aten::max_pool2d_with_indices)
 *   pai_maxpool2_bwd  dx [N][H][W][C] = dout routed to the arg-max, zeros elsewhere
 *   pai_upsample2     out [N][2H][2W][C];  pai_upsample2_bwd  dx [N][H][W][C] = sum of the 2x2 window of dout
 *   pai_add_act       out = act(a + b)   (backward: pai_act_bwd on the stored sum)
 * ------------------------------------------------------------------------- */
int pai_maxpool2(int dtype, const void* x, int N, int H, int W, int C, void* out, unsigned char* idx, void* stream);
int pai_maxpool2_bwd(int dtype, const void* dout, const unsigned char* idx, int N, int H, int W, int C, void* dx,
                     void* stream);
int pai_upsample2(int dtype, const void* x, int N, int H, int W, int C, void* out, void* stream);
int pai_upsample2_bwd(int dtype, const void* dout, int N, int H, int W, int C, void* dx, void* stream);
int pai_add_act(int dtype, const void* a, const void* b, int64_t numel, int act, void* out, void* stream);
/* The tail of a residual block, backward (reference models/res_unet.py:165-171: conv_block(x) + conv_skip(x), both ending in
 * a BatchNorm2d): the two BatchNorms read the SAME incoming gradient d.  pai_bn2_bwd_reduce = pai_bn_bwd_reduce(_affine)
 * for both branches (branch a through its activation act_a: du_a = d * act_a'(za * scale_a + shift_a), not stored; branch b
 * without one) + the two pai_bn_bwd_finalize (sums_x = [sum du | sum du * xhat], fp32 [2 C]); pai_bn2_bwd_apply = the two
 * pai_bn_bwd_apply(_affine).  Big bf16 tensors take one pass over (d, za, zb) each -- 3 + 5 tensor passes instead of
 * 4 + 6; everything else runs the one-branch calls twice.  part_x: fp32 [pai_bn_bwd_partial_rows(M)][2][C].  ABI 132. */
int pai_bn2_bwd_reduce(int dtype, const void* d, int act_a, const void* za, const void* zb, int64_t M, int C,
                       const float* scale_a, const float* shift_a, const float* mean_a, const float* rstd_a,
                       const float* mean_b, const float* rstd_b, float* part_a, float* part_b, float* sums_a, float* sums_b,
                       void* stream);
int pai_bn2_bwd_apply(int dtype, const void* d, int act_a, const void* za, const void* zb, int64_t M, int C,
                      const float* scale_a, const float* shift_a, const float* mean_a, const float* rstd_a,
                      const float* gamma_a, const float* sums_a, const float* mean_b, const float* rstd_b,
                      const float* gamma_b, const float* sums_b, void* dza, void* dzb, void* stream);
/* nn.InstanceNorm2d(C) (affine=False, no running statistics) with the activation behind it -- DiscriminatorBlock(norm=True),
 * reference models/wrapper.py:203-205; x, y: NHWC [N][HW][C], C a multiple of 8; mean, rstd: fp32 [N][C] (kept for the backward
 * pass).  y = act((x - mean[n][c]) * rstd[n][c]) with the biased variance over the HW pixels of sample n;
 * backward: du = g * act'(y), dx = rstd * (du - mean(du) - xhat * mean(du * xhat)).  ABI 132. */
int pai_instnorm_fwd(int dtype, const void* x, int N, int HW, int C, float eps, int act, void* y, float* mean, float* rstd,
                     void* stream);
int pai_instnorm_bwd(int dtype, const void* g, const void* x, int N, int HW, int C, int act, const float* mean,
                     const float* rstd, void* dx, void* stream);

/* ---------------------------------------------------------------------------
 * TransUNet (models/trans_unet.py) pieces besides its convolutions / BatchNorms / Upsample (above) and its
 * nn.Linear layers, which run as pointwise convolutions over the token rows (pai_conv_* with kernel = 1,
 * N = 1, H = 1, W = tokens).  Token tensors are [M][D] in the storage dtype, M = tokens, D = features.
 *
 *   pai_layernorm_fwd   nn.LayerNorm (:142,144; norm1 / norm2 of nn.TransformerEncoderLayer :151-156) with the
 *                       residual sum in front of it and a broadcast addend behind it fused:
 *                         s = x + res (written to sum_out; res == NULL: s = x, sum_out unused)
 *                         y = (s - mean) * rstd * gamma + beta + post[row % P]     (post fp32 [P][D] or NULL:
 *                                                            the `patch_emb += pos_embedding` of :172)
 *                       mean / rstd: fp32 [M] (biased variance, eps inside the root) for the backward pass.
 *   pai_layernorm_bwd   dx = d/ds of the above from dy (the gradient w.r.t. y; it is also the gradient of x, res
 *                       and, summed over the rows of equal row % P, of post); xs = the normalised tensor (sum_out
 *                       or x).  dgamma_dbeta (fp32 [2][D]: row 0 = dbeta, row 1 = dgamma, overwritten) or NULL;
 *                       partials: fp32 [pai_layernorm_partial_rows(M)][2][D] workspace.
 *   pai_gelu(_bwd)      erf GELU (activation = "gelu", :155) of the stored pre-activation z.
 *   pai_mha_fwd / _bwd  attention core of nn.MultiheadAttention on packed projections qkv [S*B][3E]
 *                       (row = s*B + b, columns q | k | v, head h = columns h*hd..): out [S*B][E] =
 *                       softmax(q k^T / sqrt(hd)) v per (b, h); probs fp32 [B*heads][S][S] is kept for the
 *                       backward pass, which writes dqkv [S*B][3E]; ds_workspace fp32 like probs.
 *                       mask (fp32, shaped like probs, 0 or 1 / (1 - p); NULL = none): the Dropout on the attention
 *                       weights, out = (softmax(..) * mask) v; drawing it is the caller's RNG plumbing.
 *                       batch_first = False in the reference: S is the IMAGE batch, B the patch count.
 *   pai_subsample2      out[n][y][x][c] = x[n][2y][2x][c]; _bwd writes the zero-filled adjoint [N][H][W][C].
 *                       Conv2d(k3, s2, p1) = subsample(Conv2d(k3, s1, p1)), Conv2d(k1, s2) = Conv2d(k1)(subsample)
 *                       (EncoderBlock, :203-227).
 *   pai_bn_stats        BatchNorm partial statistics [pai_bn_stats_rows(M)][2][C] of a stored tensor [M][C] (the
 *                       subsampled convolution output), for pai_bn_finalize; allocate
 *                       pai_bn_stats_buffer_rows(pai_bn_stats_rows(M)) rows.
 * ------------------------------------------------------------------------- */
int pai_layernorm_partial_rows(int64_t M);
int pai_layernorm_fwd(int dtype, const void* x, const void* res, int64_t M, int D, const float* gamma,
                      const float* beta, float eps, const float* post, int P, void* sum_out, void* y,
                      float* mean, float* rstd, void* stream);
int pai_layernorm_bwd(int dtype, const void* dy, const void* xs, int64_t M, int D, const float* gamma,
                      const float* mean, const float* rstd, void* dx, float* dgamma_dbeta, float* partials,
                      void* stream);
int pai_gelu(int dtype, const void* z, int64_t numel, void* out, void* stream);
int pai_gelu_bwd(int dtype, const void* dy, const void* z, int64_t numel, void* dz, void* stream);
int pai_mha_fwd(int dtype, const void* qkv, int S, int B, int heads, int hd, const float* mask, void* out,
                float* probs, void* stream);
int pai_mha_bwd(int dtype, const void* dout, const void* qkv, const float* probs, int S, int B, int heads,
                int hd, const float* mask, void* dqkv, float* ds_workspace, void* stream);
int pai_subsample2(int dtype, const void* x, int N, int H, int W, int C, void* out, void* stream);
int pai_subsample2_bwd(int dtype, const void* dout, int N, int H, int W, int C, void* dx, void* stream);
int pai_bn_stats_rows(int64_t M);
int pai_bn_stats(int dtype, const void* z, int64_t M, int C, float* stats, void* stream);
/* out[C] += column sums of x [rows][C] (storage dtype), e.g. the gradient of a broadcast addend. */
int pai_colsum(int dtype, const void* x, int64_t rows, int C, float* out, void* stream);

/* ---------------------------------------------------------------------------
 * Attention gate of the Attention U-Net skip connections.  Replaces the ATen ops behind
 * AttentionBlock.forward (models/attention_unet.py:88-96) that are not convolutions:
 *   h = ReLU(BN_s(sg) + BN_i(ig)),  logit = conv1x1(h; w_a, b_a)  (K -> 1),
 *   att = Sigmoid(BN_a(logit)),  out = x * att
 * and their autograd.  ig / sg are the raw outputs of the two C -> K pointwise convolutions
 * (pai_conv_fwd with kernel = 1), K = C / 2; BatchNorm statistics are finalised by
 * pai_bn_finalize (C = K for BN_i / BN_s, C = 1 for BN_a) from the partial rows written here.
 * Tensors are NHWC [M][channels] in the storage dtype; logit / att / dl are fp32 [M].
 * `partials`: fp32 [pai_gate_partial_rows(M)][2][1] (forward: sum, sum of squares of logit;
 * backward: sum dl, sum dl * xhat_a), sized through pai_bn_stats_buffer_rows for pai_bn_finalize.
 * ------------------------------------------------------------------------- */
int pai_gate_partial_rows(int64_t M);
int pai_gate_hidden(int dtype, const void* ig, const void* sg, int64_t M, int K, const float* scale_i,
                    const float* shift_i, const float* scale_s, const float* shift_s, const float* w_a,
                    const float* b_a, void* h, float* logit, float* partials, void* stream);
int pai_gate_apply(int dtype, const void* x, const float* logit, int64_t M, int C, const float* scale_a,
                   const float* shift_a, void* out, float* att, void* stream);
/* dx_skip = dout * att;  dl = <dout, x> * att * (1 - att)  (gradient w.r.t. BN_a's output);
 * partials rows = (sum dl, sum dl * (logit - mean_a) * rstd_a).  relu_out != 0: dout is the gradient
 * w.r.t. ReLU(out) (the nn.ReLU in front of the consuming DecoderBlock, models/pix2pix.py:98) and is
 * masked with out > 0 first. */
int pai_gate_apply_bwd(int dtype, const void* dout, const void* x, const float* att, const float* logit,
                       int64_t M, int C, const float* mean_a, const float* rstd_a, void* dx_skip, float* dl,
                       float* partials, int relu_out, void* stream);
/* BN_a backward (sums_a from pai_bn_bwd_finalize), then through w_a and the ReLU:
 *   dsum = 1[h > 0] * dlogit * w_a  (gradient w.r.t. both BN_i's and BN_s's output),
 *   dw_a += sum dlogit * h,  db_a += sum dlogit,
 *   partials_i / partials_s rows = (sum dsum, sum dsum * xhat_i|s)  [pai_gate_partial_rows(M)][2][K]. */
int pai_gate_hidden_bwd(int dtype, const float* dl, const float* logit, const void* h, const void* ig,
                        const void* sg, int64_t M, int K, const float* mean_a, const float* rstd_a,
                        const float* gamma_a, const float* sums_a, const float* w_a, const float* mean_i,
                        const float* rstd_i, const float* mean_s, const float* rstd_s, void* dsum,
                        float* partials_i, float* partials_s, float* dw_a, float* db_a, void* stream);

/* ---------------------------------------------------------------------------
 * Losses.  Replace F.binary_cross_entropy_with_logits / F.l1_loss / F.mse_loss
 * at models/wrapper.py:45-49,66,84-93.  Each accumulates
 * `loss_scale * mean-reduced loss` into *loss (fp64 device scalar, += ) and,
 * when grad != NULL, writes grad = grad_scale * d(mean loss)/d(input).
 * ------------------------------------------------------------------------- */
int pai_bce_logits(const float* logits, int64_t numel, float target, float loss_scale,
                   double* loss, float grad_scale, float* grad, void* stream);
int pai_l1(const float* pred, const float* target, int64_t numel, float loss_scale, double* loss,
           float grad_scale, float* grad, void* stream);
int pai_mse(const float* pred, const float* target, int64_t numel, float loss_scale, double* loss,
            float grad_scale, float* grad, void* stream);
/* *out = (float)*acc; *acc = 0.  The fp64 accumulator a group of loss launches added into becomes the fp32 loss value
 * and is re-armed for the next step in one single-thread launch (the reference sums fp32 scalars with tensor ops,
 * models/wrapper.py:50,94). */
int pai_scalar_take(double* acc, float* out, void* stream);
/* sums = {sum of per-image SSIM, sum of squared errors} as pai_ssim_sse accumulates them ->
 * out3 = {mean SSIM, PSNR = -10 log10(mse), RMSE = sqrt(mse)} (models/utils.py:38-47), sums re-armed to 0. */
int pai_metrics_take(double* sums, int64_t n_images, int64_t numel, float* out3, void* stream);
/* Generator head backward: dh = (g_disc + g_rec) * (1 - pred^2)  (tanh', models/pix2pix.py:216);
 * g_disc / g_rec fp32 or NULL; dh in storage dtype. */
int pai_tanh_bwd(int dtype, const float* pred, const float* g_a, const float* g_b, int64_t numel,
                 void* dh, void* stream);

/* denormalize (models/utils.py:11): out = clamp(x*0.5+0.5, 0, 1).  With grad_out != NULL
 * the call computes the backward instead: out = grad_out * 0.5 where 0 <= x*0.5+0.5 <= 1. */
int pai_denormalize(const float* x, const float* grad_out_or_null, int64_t numel, float* out,
                    void* stream);

/* ---------------------------------------------------------------------------
 * Metrics.  Replace torchmetrics.functional SSIM / PSNR / MSE as called from
 * models/utils.py:38-47 and report.py:78-96 on denormalised images
 * (models/utils.py:11: clamp(x*0.5+0.5, 0, 1); `denorm` != 0 fuses it).
 * All reductions accumulate (+=) in fp64; the caller zeroes them.
 * out2[0] += sum over images (N*C planes) of the per-plane 5-px-cropped SSIM mean
 * out2[1] += sum of squared error over all pixels
 * per_image (fp64 [N*C], +=) and full_map (fp32 [N*C][H][W], un-cropped) optional.
 * ------------------------------------------------------------------------- */
int pai_ssim_sse(const float* pred, const float* target, int NC, int H, int W, int denorm,
                 double* out2, double* per_image, float* full_map, void* stream);
/* Gradient of  -(w_ssim * mean SSIM + w_psnr * PSNR)  w.r.t. the (un-denormalised)
 * prediction, for loss_type ssim / psnr / ssim+psnr (models/wrapper.py:53-63).
 * sse: device pointer to the sum of squared error (out2[1] above). */
int pai_ssim_psnr_bwd(const float* pred, const float* target, int NC, int H, int W, int denorm,
                      float w_ssim, float w_psnr, const double* sse, float* grad, float* workspace,
                      void* stream);
int64_t pai_ssim_bwd_workspace_floats(int NC, int H, int W);

/* ---------------------------------------------------------------------------
 * Utilities
 * ------------------------------------------------------------------------- */
int pai_cast(int src_dtype, const void* src, int dst_dtype, void* dst, int64_t numel, void* stream);
/* out[C] (+)= sum over rows of partial[rows][C], accumulated in fp64. */
int pai_reduce_rows(const float* partial, int rows, int C, float* out, int accumulate, void* stream);
/* Fused Adam over one flat fp32 arena (torch.optim.Adam semantics, no weight
 * decay / amsgrad; models/wrapper.py:98-111).  step_count is the 1-based step. */
int pai_adam(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t numel,
             float lr, float beta1, float beta2, float eps, int step_count, void* stream);
/* pai_adam over a range of the arena that holds ONE dense conv weight in master order [Cout][taps][Cin] (Cin, Cout
 * multiples of 64) at element offset w_off, and writes that weight's bf16 filter packs (as pai_pack_weights would: w_fwd
 * in master order, w_dgrad [Cin][taps][Cout]; either may be NULL) from the block that produced the new values.  The
 * rest of the range (bias, BatchNorm affine parameters) gets the plain update.  Same result, bit for bit, as pai_adam
 * followed by pai_pack_weights.  All pointers 16-byte aligned, w_off a multiple of 4. */
int pai_adam_pack(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t numel, int64_t w_off,
                  int Cout, int taps, int Cin, void* w_fwd, void* w_dgrad, float lr, float beta1, float beta2, float eps,
                  int step_count, void* stream);
/* pai_adam with the step count in device memory (*step_dev is advanced by the call; coeff2_dev: two floats of device
 * scratch): what a training step captured into a hipGraph needs -- a host-side step count would be frozen into the
 * graph and every replay would reuse the captured step's bias correction. */
int pai_adam_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t numel,
                 float lr, float beta1, float beta2, float eps, int64_t* step_dev, float* coeff2_dev, void* stream);
/* The same update over `count` separately allocated fp32 tensors (HOST arrays of device pointers and element
 * counts), a few launches in all: the optimizer of the composable networks (models/res_unet.py, models/trans_unet.py),
 * whose parameters are not one arena. */
int pai_adam_multi(int count, void* const* params, const void* const* grads, void* const* exp_avgs,
                   void* const* exp_avg_sqs, const int64_t* numels, float lr, float beta1, float beta2, float eps,
                   int step_count, void* stream);
/* ... with the step count in device memory (a captured step, see pai_adam_dev): *step_dev is advanced ONCE per call. */
int pai_adam_multi_dev(int count, void* const* params, const void* const* grads, void* const* exp_avgs,
                       void* const* exp_avg_sqs, const int64_t* numels, float lr, float beta1, float beta2, float eps,
                       int64_t* step_dev, float* coeff2_dev, void* stream);

/* ---------------------------------------------------------------------------
 * Gradient exchange (data parallelism, one process per GPU; replaces the all-reduce Lightning's DDP wrapper issues
 * for the reference, main.py:123-136): in-place SUM all-reduce of a device buffer over RCCL / xGMI on the caller's
 * stream.  librccl.so is opened with dlopen on first use (PAI_RCCL_LIB overrides the name).
 *   pai_comm_unique_id  rank 0 fills 128 bytes (host memory); the caller ships them to the other ranks
 *   pai_comm_init       collective over all ranks of the job; the calling thread's current HIP device is the rank's GPU
 *   pai_allreduce       dtype PAI_F32 | PAI_BF16; asynchronous, ordered on `stream`; averaging is the caller's scale
 * ------------------------------------------------------------------------- */
#define PAI_COMM_ID_BYTES 128
int pai_comm_unique_id(void* id_out_host);
int pai_comm_init(const void* id_host, int rank, int world, void** comm_out);
int pai_allreduce(void* comm, void* ptr, int64_t count, int dtype, void* stream);
int pai_comm_destroy(void* comm);

/* ---------------------------------------------------------------------------
 * Launch plans: the training step as ONE host call.
 * The reference's step is a single Python call (models/wrapper.py:117-162: UnetWrapper.training_step); issued launch
 * by launch through this ABI it is ~205 calls and 5.5 ms of host time per 6.4 ms step.  A plan records the launches the
 * library makes between pai_plan_begin and pai_plan_end -- kernel, grid, LDS size, stream and the argument block BY
 * VALUE -- while they execute normally, from every host thread of the process (autograd runs the backward pass on a
 * thread of its own); pai_plan_run re-issues the sequence on the same streams in the same order.
 *   - the caller owns validity: every device pointer a recorded call was given must still be alive and mean the same
 *     thing at replay (activations, packs, arenas, the handle's workspaces).  Re-record after any buffer changes.
 *   - cross-stream ordering must go through pai_stream_wait to be part of the plan (an event recorded on
 *     `signalling_stream`, waited for by `waiting_stream`; outside a recording it is the same edge, eagerly).
 *   - pai_adam / pai_adam_pack / pai_adam_multi take the optimizer step count t by value: a recorded launch keeps t0 and
 *     replays with t0 + step_delta (the number of optimizer steps taken since the recording), bit-identical to the
 *     eager launch of that step.  Everything else step-dependent already lives on the device.
 *   - pai_allreduce calls are recorded too (the data-parallel step as one plan per rank).
 *   - one plan may be recorded at a time per process; replay is not re-entrant per plan.
 *   - while a plan is being recorded EVERY launch of the library in the process is appended, whichever thread or device
 *     made it: keep unrelated users of the library (a metrics / validation thread, a second model, another in-process
 *     rank) quiet between pai_plan_begin and pai_plan_end.  pai_plan_run must be called with the device current on
 *     which the plan was recorded (checked since ABI 131).
 * pai_plan_info: launches = kernel + memset + collective nodes, waits = stream-wait edges, streams = distinct streams.
 * ------------------------------------------------------------------------- */
typedef struct pai_plan_s* pai_plan_t;
int pai_plan_create(pai_plan_t* out);
int pai_plan_destroy(pai_plan_t plan);
int pai_plan_begin(pai_plan_t plan);
int pai_plan_end(pai_plan_t plan);
int pai_plan_run(pai_plan_t plan, int64_t step_delta);
int pai_plan_info(pai_plan_t plan, int* launches, int* waits, int* streams, int64_t* runs);
int pai_stream_wait(void* waiting_stream, void* signalling_stream);
/* The same edge when its source is the LAST launch this library made on `signalling_stream` and the caller enqueued
 * nothing else there since.  Executed eagerly it is pai_stream_wait; in a plan the source launch itself carries the
 * event, so the replayed signalling stream has no marker packet between that launch and the next one. */
int pai_stream_wait_last(void* waiting_stream, void* signalling_stream);
/* The same edge in two halves, for a wait that is issued later than the point it refers to (the thin weight gradients
 * of one backward pass share scratch: the second waits for the mark behind the first, not for what the first one's
 * stream was given since).  Caller-owned events (no timing); a recorded plan keeps the handle, so the event must outlive
 * every plan that was recorded while it was in use. */
typedef struct pai_event_s* pai_event_t;
int pai_event_create(pai_event_t* out);
int pai_event_destroy(pai_event_t ev);
int pai_event_record(pai_event_t ev, void* stream);
/* Measurement: events WITH timing, and "the next launch this thread makes through the library carries (start, stop) as its
 * own start / stop events" -- the launch's duration as the command processor stamps it (what rocprofv3 --kernel-trace
 * reports), without marker packets around it.  pai_profile_arm(NULL, NULL) disarms.  bench.py's roofline uses it. */
int pai_event_create_timing(pai_event_t* out);
int pai_event_elapsed_ms(pai_event_t start, pai_event_t stop, float* ms_host);
int pai_profile_arm(pai_event_t start, pai_event_t stop);
int pai_stream_wait_event(void* waiting_stream, pai_event_t ev);
/* p[i][0 .. numel[i]) = 0 for `count` fp32 buffers in one launch per 96 buffers (host pointer tables): the accumulated
 * segments of a gradient arena in front of a backward pass (replaces torch._foreach_zero_, so that the clear is a node
 * of the plan like everything else). */
int pai_zero_multi(int count, void* const* ptrs, const int64_t* numels, void* stream);
/* Up to 8 pai_cast calls of one dtype pair in ONE launch (host pointer tables; every numel a multiple of 8, every pointer
 * 16-byte aligned): the four NHWC input copies in front of the batched PatchGAN pass (models/wrapper.py:236-238, torch.cat
 * of the conditioning image and the real / generated one) were four 7-12 us launches. */
int pai_cast_multi(int count, int src_dtype, const void* const* srcs, int dst_dtype, void* const* dsts,
                   const int64_t* numels, void* stream);
/* dst[i][k] += weight * (src[i][k] - dst[i][k]) for `count` pairs of fp32 buffers (host pointer tables), one launch per 48
 * pairs: the exponential-moving-average update the reference runs after every training batch over ALL module parameters
 * (callbacks/ema.py:24-33 -> torch_ema.ExponentialMovingAverage.update: shadow -= (1 - decay) * (shadow - param)).
 * Written as that subtraction (s - w * (s - p), one rounding per operation, no contraction), so that it matches torch_ema's
 * in-place sub_ / mul_ sequence bit for bit.  Callers pass whole arena ranges where parameters are adjacent in memory (one
 * pair per network then); 16-byte alignment is not required. */
int pai_lerp_multi(int count, void* const* dsts, const void* const* srcs, const int64_t* numels, float weight, void* stream);

/* nn.Conv2d filters ([Cout][Cin / groups][kh][kw] fp32; reference models/res_unet.py:147-151 uses groups = 32) to and from
 * the dense tap-major fp32 layout [Cout][kh * kw][Cin] of pai_pack_weights / pai_conv_wgrad (groups as diagonal blocks,
 * zeros elsewhere): one launch each on the caller's stream. */
int pai_filter_to_dense(const float* w_oihw, int Cout, int Cin_per_group, int taps, int groups, float* dense, void* stream);
int pai_filter_grad_from_dense(const float* dense_dw, int Cout, int Cin_per_group, int taps, int groups, float* dw_oihw,
                               void* stream);

/* dst[a][c][b][:] = src[a][b][c][:] for a contiguous [A][B][C][D] tensor of 2- or 4-byte elements (D * elem_bytes a
 * multiple of 16): the "n c (h p1) (w p2) <-> n (h w) (p1 p2 c)" rearrangement around the ViT bottleneck (reference
 * models/trans_unet.py:139-141,175-179) on NHWC storage, with A = n * grid, (B, C) = (p1, grid) or (grid, p1), D = p2 * c. */
int pai_swap_mid(int elem_bytes, const void* src, int64_t A, int B, int C, int64_t D, void* dst, void* stream);
/* ptr[0 .. numel) *= factor (fp32, 16-byte aligned): the x 1/world_size average behind the SUM all-reduce of a gradient
 * bucket (DDP averages, reference main.py:123-136 through pl.Trainer), as a node of the plan. */
int pai_scale(float* ptr, int64_t numel, float factor, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PAI_HIP_H */

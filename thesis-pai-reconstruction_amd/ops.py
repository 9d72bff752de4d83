"""Thin tensor-level wrappers over the C ABI (one Python function per entry point).

Tensors only supply device memory and the stream; all arithmetic happens in
libpai_hip.so.  Every wrapper insists on CUDA(HIP) tensors -- there is no CPU path.
"""
from __future__ import annotations

import ctypes as C

import threading

import torch

from . import lib as L
from .lib import ACT_LRELU, ACT_NONE, ACT_RELU, ACT_TANH, BF16, F32, ConvDesc, PaiError  # noqa: F401


def code_of(dtype: torch.dtype) -> int:
    if dtype == torch.float32:
        return F32
    if dtype == torch.bfloat16:
        return BF16
    raise PaiError(f"unsupported storage dtype {dtype}")


_TLS = threading.local()


def _stream():
    s = getattr(_TLS, "stream", None)
    return s if s is not None else torch.cuda.current_stream().cuda_stream


class on_stream:
    """Library launches of this thread go to `stream` inside the block; torch's current stream is NOT switched (allocations
    and tensor-library kernels stay where they were) -- a few hundred ns instead of the ~20 us of ``torch.cuda.stream``."""

    def __init__(self, stream):
        self.raw = stream.cuda_stream

    def __enter__(self):
        self.prev = getattr(_TLS, "stream", None)
        _TLS.stream = self.raw

    def __exit__(self, *exc):
        _TLS.stream = self.prev
        return False


def _p(t, dtype=None):
    if t is None:
        return None
    if not t.is_cuda:
        raise PaiError("pai ops need HIP device tensors (no CPU fallback exists)")
    if dtype is not None and t.dtype != dtype:
        raise PaiError(f"expected {dtype}, got {t.dtype}")
    return t.data_ptr()


def make_desc(dtype, transposed, N, H, W, C1, C2, Cout, stride=2, relu1=0, relu2=0, act=ACT_NONE,
              kernel=4, groups=1) -> ConvDesc:
    d = ConvDesc()
    d.dtype = code_of(dtype)
    d.transposed = int(transposed)
    d.N, d.H, d.W, d.C1, d.C2, d.Cout = N, H, W, C1, C2, Cout
    d.kernel, d.stride, d.pad = {4: (4, stride, 1), 3: (3, 1, 1), 1: (1, 1, 0)}[kernel]
    d.relu1, d.relu2, d.epilogue_act = int(relu1), int(relu2), int(act)
    d.groups = int(groups) if groups and groups > 1 else 0
    return d


def conv_out_hw(d: ConvDesc):
    oh, ow = C.c_int(), C.c_int()
    L.check(L.load().pai_conv_out_hw(C.byref(d), C.byref(oh), C.byref(ow)), "pai_conv_out_hw")
    return oh.value, ow.value


def conv_fwd_stats_rows(d: ConvDesc) -> int:
    r = L.load().pai_conv_fwd_stats_rows(C.byref(d))
    if r < 0:
        L.check(1, "pai_conv_fwd_stats_rows")
    return r


def conv_fwd_stats_rows_max(d: ConvDesc) -> int:
    return L.load().pai_conv_fwd_stats_rows_max(C.byref(d))


def bn_stats_buffer_rows(rows: int) -> int:
    return L.load().pai_bn_stats_buffer_rows(rows)


# ---- optional per-launch timing (bench.py roofline accounting) -------------------------------
# When PROFILE is a list, every convolution-family call is bracketed by HIP events recorded on
# the stream the kernel is launched on, and (kernel family, op, algorithmic FLOPs, events) is
# appended.  Off (None) in normal operation.
PROFILE = None
KERNEL_NAMES = {0: "gg_simt", 1: "gg_rowdot", 2: "gg_mfma_bf16_t128", 3: "gg_mfma_bf16_t64", 4: "thin_mfma_bf16", 5: "small_mfma_bf16", 6: "grouped3_k",
                7: "pwx_k"}


def conv_kernel_id(d: ConvDesc, op: int) -> int:
    return L.load().pai_conv_kernel_id(C.byref(d), op)


def conv_kernel_name(d: ConvDesc, op: int) -> str:
    """rocprofv3 symbol of the main kernel the call launches (family name for the non-MFMA families)."""
    buf = C.create_string_buffer(96)
    if L.load().pai_conv_kernel_name(C.byref(d), op, buf, 96) != 0:
        L.check(1, "pai_conv_kernel_name")
    return buf.value.decode()


def conv_flops(d: ConvDesc) -> int:
    """Algorithmic FLOPs (2 x MAC, padding taps included) of one fwd / dgrad / wgrad launch; a grouped convolution
    (pai_conv_desc.groups) counts the MACs of its groups only, not the structural zeros of the block-diagonal form."""
    groups = max(1, int(getattr(d, "groups", 1)))
    if d.transposed:
        return 2 * d.N * d.H * d.W * 16 * (d.C1 + d.C2) * d.Cout // groups
    oh = (d.H + 2 * d.pad - d.kernel) // d.stride + 1
    ow = (d.W + 2 * d.pad - d.kernel) // d.stride + 1
    return 2 * d.N * oh * ow * d.kernel * d.kernel * (d.C1 + d.C2) * d.Cout // groups


class LaunchTimer:
    """start / stop timing events that ride on ONE launch (pai_profile_arm -> hipExtLaunchKernel): ``elapsed_time`` is the
    kernel's own duration, as rocprofv3 reports it.  Events recorded around a launch put marker packets on the stream and
    read 10-15 % long (round 3: 153.8 us under brackets against 134.6 us in the kernel trace)."""

    def __init__(self):
        self.h0, self.h1 = C.c_void_p(), C.c_void_p()
        L.check(L.load().pai_event_create_timing(C.byref(self.h0)), "pai_event_create_timing")
        L.check(L.load().pai_event_create_timing(C.byref(self.h1)), "pai_event_create_timing")

    def arm(self):
        L.check(L.load().pai_profile_arm(self.h0, self.h1), "pai_profile_arm")

    @staticmethod
    def disarm():
        L.check(L.load().pai_profile_arm(None, None), "pai_profile_arm")

    def elapsed_time(self, _other=None) -> float:
        ms = C.c_float()
        L.check(L.load().pai_event_elapsed_ms(self.h0, self.h1, C.byref(ms)), "pai_event_elapsed_ms")
        return ms.value

    def __del__(self):
        try:
            L.load().pai_event_destroy(self.h0)
            L.load().pai_event_destroy(self.h1)
        except Exception:
            pass


class _Timed:
    """The FIRST launch of the bracketed call (the convolution-family kernel itself; finish / BatchNorm launches behind it
    are not part of the figure) is timed by events of its own."""

    def __init__(self, d, op):
        self.on = PROFILE is not None
        if self.on:
            self.d, self.op = d, op
            self.t = LaunchTimer()

    def __enter__(self):
        if self.on:
            self.t.arm()

    def __exit__(self, *exc):
        if self.on:
            LaunchTimer.disarm()
            # forward and input-gradient share kernels; keyed by the symbol rocprofv3 reports
            PROFILE.append((conv_kernel_name(self.d, self.op), self.op, conv_flops(self.d), self.t, self.t))
        return False


# The HBM-bound passes (BatchNorm apply / backward reduce / backward apply, Adam): when PROFILE_HBM is a list every call
# is bracketed the same way and (pass family, algorithmic bytes, events) is appended -- bench.py's `roofline_hbm`.
PROFILE_HBM = None


class _TimedBytes:
    def __init__(self, family, nbytes):
        self.on = PROFILE_HBM is not None
        if self.on:
            self.family, self.nbytes = family, int(nbytes)
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)

    def __enter__(self):
        if self.on:
            self.e0.record()

    def __exit__(self, *exc):
        if self.on:
            self.e1.record()
            PROFILE_HBM.append((self.family, self.nbytes, self.e0, self.e1))
        return False


def _es(dtype) -> int:
    return 4 if dtype == torch.float32 else 2


class Handle:
    """A per-device handle of libpai_hip.so (pai_create / pai_bind / pai_destroy, include/pai_hip.h) together with the
    torch tensors registered as its split-K workspace and scratch (the library never allocates).  One per device is
    created on first use (``handle_for``); more can be made for independent model replicas and switched with
    ``bind()``."""

    def __init__(self, device):
        dev = torch.device(device)
        self.device = torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())
        self._h = C.c_void_p()
        L.check(L.load().pai_create(self.device.index, C.byref(self._h)), "pai_create")
        self.workspace = None
        self.scratch = None
        self.wgrad_workspace = None
        # bumped whenever a registered buffer is replaced: a caller's captured hipGraph or a recorded launch
        # plan holds the OLD addresses in its kernel arguments and must not be replayed past such a change
        self.epoch = 0

    def bind(self):
        L.check(L.load().pai_bind(self._h), "pai_bind")
        return self

    def ensure_workspace(self, nbytes: int) -> None:
        """Register (grow) the zero-filled split-K workspace (pai_handle_set_workspace)."""
        if nbytes <= 0:
            return
        if self.workspace is None or self.workspace.numel() * 4 < nbytes:
            self.workspace = torch.zeros((nbytes + 3) // 4, dtype=torch.float32, device=self.device)
            self.epoch += 1
            L.check(L.load().pai_handle_set_workspace(self._h, self.workspace.data_ptr(), self.workspace.numel() * 4),
                    "pai_handle_set_workspace")

    def ensure_scratch(self, nbytes: int) -> None:
        """Register (grow) the general scratch buffer (pai_handle_set_scratch)."""
        if nbytes <= 0:
            return
        if self.scratch is None or self.scratch.numel() * 4 < nbytes:
            self.scratch = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=self.device)
            self.epoch += 1
            L.check(L.load().pai_handle_set_scratch(self._h, self.scratch.data_ptr(), self.scratch.numel() * 4),
                    "pai_handle_set_scratch")

    def ensure_wgrad_workspace(self, nbytes: int) -> None:
        """Register (grow) the weight-gradient slab buffer (pai_handle_set_wgrad_workspace)."""
        if nbytes <= 0:
            return
        if self.wgrad_workspace is None or self.wgrad_workspace.numel() * 4 < nbytes:
            if self.wgrad_workspace is not None:
                torch.cuda.synchronize(self.device)     # side-stream launches may still be writing the old buffer
            self.wgrad_workspace = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=self.device)
            self.epoch += 1
            L.check(L.load().pai_handle_set_wgrad_workspace(self._h, self.wgrad_workspace.data_ptr(),
                                                           self.wgrad_workspace.numel() * 4),
                    "pai_handle_set_wgrad_workspace")

    def close(self):
        if self._h:
            L.check(L.load().pai_destroy(self._h), "pai_destroy")
            self._h = C.c_void_p()
            _HANDLES.pop(self.device.index, None) if _HANDLES.get(self.device.index) is self else None


_HANDLES = {}


def handle_for(device) -> Handle:
    """The default handle of ``device`` (created, and therefore active, on first use)."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    h = _HANDLES.get(idx)
    if h is None:
        h = _HANDLES[idx] = Handle(torch.device("cuda", idx))
    return h


def conv_workspace_bytes(d: ConvDesc, op: int) -> int:
    return L.load().pai_conv_workspace_bytes(C.byref(d), op)


def ensure_workspace(nbytes: int, device) -> None:
    """Grow the split-K workspace of the device's default handle."""
    handle_for(device).ensure_workspace(nbytes)


def conv_scratch_bytes(d: ConvDesc, op: int) -> int:
    return L.load().pai_conv_scratch_bytes(C.byref(d), op)


def scratch_bytes_for(descs) -> int:
    """Scratch to register for a set of layers: forward / input-gradient calls use the head of the buffer, the
    weight-gradient calls (which run on a second stream) its tail -- see the stream contract in include/pai_hip.h."""
    head = max((conv_scratch_bytes(d, op) for d in descs for op in (0, 1)), default=0)
    tail = max((conv_scratch_bytes(d, 2) for d in descs), default=0)
    return head + tail


def ensure_scratch(nbytes: int, device) -> None:
    """Grow the general scratch buffer of the device's default handle."""
    handle_for(device).ensure_scratch(nbytes)


def conv_wgrad_workspace_bytes(d: ConvDesc) -> int:
    return L.load().pai_conv_wgrad_workspace_bytes(C.byref(d))


def ensure_wgrad_workspace(descs, device) -> None:
    """Grow the weight-gradient slab buffer of the device's default handle to what the layers `descs` need."""
    handle_for(device).ensure_wgrad_workspace(max((conv_wgrad_workspace_bytes(d) for d in descs), default=0))


def conv_fwd(d, x1, x2, w, bias, y_raw=None, y_act=None, y_f32=None, stats=None):
    with _Timed(d, 0):
        L.check(L.load().pai_conv_fwd(C.byref(d), _p(x1), _p(x2), _p(w), _p(bias, torch.float32), _p(y_raw),
                                      _p(y_act), _p(y_f32, torch.float32), _p(stats, torch.float32),
                                      _stream()), "pai_conv_fwd")


def conv_prologue_ok(d) -> bool:
    """Forward and weight gradient of this layer can read their input through a prologue (pai_conv_prologue_ok)."""
    return bool(L.load().pai_conv_prologue_ok(C.byref(d)))


def conv_fwd_pro(d, x1, w, bias, y_raw, stats, pre_scale, pre_shift, pre_act):
    """y_raw = conv(act(x1 * pre_scale + pre_shift)): the producer's BatchNorm + activation applied on load."""
    with _Timed(d, 0):
        L.check(L.load().pai_conv_fwd_pro(C.byref(d), _p(x1), _p(w), _p(bias, torch.float32), _p(y_raw),
                                          _p(stats, torch.float32), _p(pre_scale, torch.float32), _p(pre_shift, torch.float32),
                                          int(pre_act), _stream()), "pai_conv_fwd_pro")


def conv_wgrad_pro(d, x1, dy, dw, dbias, overwrite, pre_scale, pre_shift, pre_act):
    with _Timed(d, 2):
        L.check(L.load().pai_conv_wgrad_pro(C.byref(d), _p(x1), _p(dy), _p(dw, torch.float32), _p(dbias, torch.float32),
                                            int(overwrite), _p(pre_scale, torch.float32), _p(pre_shift, torch.float32),
                                            int(pre_act), _stream()), "pai_conv_wgrad_pro")


def conv_dgrad(d, dy, w_dgrad, dx1, dx2=None, only_c2=False):
    with _Timed(d, 1):
        L.check(L.load().pai_conv_dgrad(C.byref(d), _p(dy), _p(w_dgrad), _p(dx1), _p(dx2), int(only_c2),
                                        _stream()), "pai_conv_dgrad")


def conv_dgrad_act(d, dy, w_dgrad, dx1, dx2, a1, act1):
    """dx1 = act1'(a1) * dgrad[:, :C1]  (pai_conv_dgrad + pai_act_bwd in one pass)."""
    with _Timed(d, 1):
        L.check(L.load().pai_conv_dgrad_act(C.byref(d), _p(dy), _p(w_dgrad), _p(dx1), _p(dx2), _p(a1), int(act1),
                                            _stream()), "pai_conv_dgrad_act")


def conv_dgrad_bn_rows_max(d) -> int:
    return L.load().pai_conv_dgrad_bn_rows_max(C.byref(d))


def conv_dgrad_bn(d, dy, w_dgrad, dx1, dx2, z, act1, add=None, act2=ACT_NONE, scale=None, shift=None,
                  mean=None, rstd=None, partials=None) -> int:
    """Input gradient with the producer's activation / BatchNorm backward (first pass) in its store
    (pai_conv_dgrad_bn).  Returns the number of partial rows written (0 without `partials`)."""
    e = L.BwdEpilogue(_p(z), _p(add), _p(scale, torch.float32), _p(shift, torch.float32), _p(mean, torch.float32),
                      _p(rstd, torch.float32), _p(partials, torch.float32), int(act1), int(act2))
    rows = C.c_int(0)
    with _Timed(d, 1):
        L.check(L.load().pai_conv_dgrad_bn(C.byref(d), _p(dy), _p(w_dgrad), _p(dx1), _p(dx2), C.byref(e),
                                           C.byref(rows), _stream()), "pai_conv_dgrad_bn")
    return rows.value


def conv_fwd_bn(d, x1, x2, w, bias, z, a, act, gamma, beta, eps, momentum, n_updates, running_mean, running_var,
                num_batches_tracked, mean, rstd, scale, shift, stats):
    """Convolution + BatchNorm2d(train) + activation in one call (pai_conv_fwd_bn)."""
    f32 = torch.float32
    bn = L.BnTrain(_p(gamma, f32), _p(beta, f32), float(eps), float(momentum), int(n_updates), _p(running_mean, f32),
                   _p(running_var, f32), _p(num_batches_tracked, torch.int64), _p(mean, f32), _p(rstd, f32),
                   _p(scale, f32), _p(shift, f32))
    with _Timed(d, 0):
        L.check(L.load().pai_conv_fwd_bn(C.byref(d), _p(x1), _p(x2), _p(w), _p(bias, f32), _p(z), _p(a), int(act),
                                         C.byref(bn), _p(stats, f32), _stream()), "pai_conv_fwd_bn")


def conv_bn_fused(d, op: int) -> bool:
    return L.load().pai_conv_bn_fused(C.byref(d), op) == 1


def conv_dgrad_bn_apply(d, dy, w_dgrad, du_scratch, dx2, z, act1, add, act2, scale, shift, mean, rstd, partials, gamma,
                        sums, dgamma, dbeta, dz):
    """Input gradient + the producer's complete BatchNorm backward (pai_conv_dgrad_bn_apply): writes dz, sums and adds
    dgamma / dbeta."""
    f32 = torch.float32
    e = L.BwdEpilogue(_p(z), _p(add), _p(scale, f32), _p(shift, f32), _p(mean, f32), _p(rstd, f32), _p(partials, f32),
                      int(act1), int(act2))
    with _Timed(d, 1):
        L.check(L.load().pai_conv_dgrad_bn_apply(C.byref(d), _p(dy), _p(w_dgrad), _p(du_scratch), _p(dx2), C.byref(e),
                                                 _p(gamma, f32), _p(sums, f32), _p(dgamma, f32), _p(dbeta, f32), _p(dz),
                                                 _stream()), "pai_conv_dgrad_bn_apply")


def bn_bwd_finalize(partials, rows, C_, sums, dgamma, dbeta):
    L.check(L.load().pai_bn_bwd_finalize(_p(partials, torch.float32), rows, C_, _p(sums, torch.float32),
                                         _p(dgamma, torch.float32), _p(dbeta, torch.float32), _stream()),
            "pai_bn_bwd_finalize")


def conv_wgrad(d, x1, x2, dy, dw, dbias=None):
    with _Timed(d, 2):
        L.check(L.load().pai_conv_wgrad(C.byref(d), _p(x1), _p(x2), _p(dy), _p(dw, torch.float32),
                                        _p(dbias, torch.float32), _stream()), "pai_conv_wgrad")


def conv_wgrad_overwrite(d, x1, x2, dy, dw, dbias=None):
    """dw = conv_backward_weight(...), dbias = ... into uninitialised buffers (pai_conv_wgrad_overwrite)."""
    with _Timed(d, 2):
        L.check(L.load().pai_conv_wgrad_overwrite(C.byref(d), _p(x1), _p(x2), _p(dy), _p(dw, torch.float32),
                                                  _p(dbias, torch.float32), _stream()), "pai_conv_wgrad_overwrite")


def conv_wgrad_overwrite_w(d, x1, x2, dy, dw, dbias=None):
    """dw = conv_backward_weight(...) into an uninitialised buffer, dbias += ... (pai_conv_wgrad_overwrite_w)."""
    with _Timed(d, 2):
        L.check(L.load().pai_conv_wgrad_overwrite_w(C.byref(d), _p(x1), _p(x2), _p(dy), _p(dw, torch.float32),
                                                    _p(dbias, torch.float32), _stream()), "pai_conv_wgrad_overwrite_w")


def pack_weights(dtype, w_master, Cout, taps, Cin, w_fwd=None, w_dgrad=None):
    L.check(L.load().pai_pack_weights(code_of(dtype), _p(w_master, torch.float32), Cout, taps, Cin, _p(w_fwd),
                                      _p(w_dgrad), _stream()), "pai_pack_weights")


TUNABLE_UNSET = -2147483648


def set_tunable(name: str, value: int = TUNABLE_UNSET) -> None:
    """Kernel-selection switch of the library (pai_set_tunable); without a value: back to the built-in default."""
    L.check(L.load().pai_set_tunable(name.encode(), int(value)), "pai_set_tunable")


def pack_weights_multi(items):
    """items: [(w_master fp32, Cout, taps, Cin, w_fwd | None, w_dgrad | None)], bf16 packs, Cin and Cout multiples of
    64: every layer in one launch."""
    import ctypes as C
    n = len(items)
    if n == 0:
        return
    ptrs = lambda k: (C.c_void_p * n)(*[_p(it[k]) if k == 0 else (_p(it[k]) if it[k] is not None else None) for it in items])
    ints = lambda k: (C.c_int32 * n)(*[int(it[k]) for it in items])
    for it in items:
        if it[0].dtype != torch.float32:
            raise PaiError("pack_weights_multi: fp32 master weights expected")
    L.check(L.load().pai_pack_weights_multi(n, ptrs(0), ints(1), ints(2), ints(3), ptrs(4), ptrs(5), _stream()),
            "pai_pack_weights_multi")


def bn_finalize(stats, rows, C_, count, gamma, beta, eps, momentum, n_updates, running_mean, running_var,
                nbt, mean, rstd, scale, shift):
    L.check(L.load().pai_bn_finalize(_p(stats), rows, C_, count, _p(gamma), _p(beta), eps, momentum, n_updates,
                                     _p(running_mean), _p(running_var), _p(nbt, torch.int64), _p(mean),
                                     _p(rstd), _p(scale), _p(shift), _stream()), "pai_bn_finalize")


def bn_eval_coeffs(C_, gamma, beta, running_mean, running_var, eps, scale, shift):
    L.check(L.load().pai_bn_eval_coeffs(C_, _p(gamma), _p(beta), _p(running_mean), _p(running_var), eps,
                                        _p(scale), _p(shift), _stream()), "pai_bn_eval_coeffs")


def bn_apply(dtype, z, M, C_, scale, shift, act, out):
    with _TimedBytes("bn_passes", 2 * M * C_ * _es(dtype)):     # read z, write the activation
        L.check(L.load().pai_bn_apply(code_of(dtype), _p(z), M, C_, _p(scale), _p(shift), act, _p(out), _stream()),
                "pai_bn_apply")


def bn2_add_act(dtype, za, scale_a, shift_a, zb, scale_b, shift_b, M, C_, act_a, act, out):
    """out = act(act_a(BN_a(za)) + BN_b(zb)) (scale_b None: + zb) in one pass (pai_bn2_add_act)."""
    with _TimedBytes("bn_passes", 3 * M * C_ * _es(dtype)):
        L.check(L.load().pai_bn2_add_act(code_of(dtype), _p(za), _p(scale_a, torch.float32), _p(shift_a, torch.float32), _p(zb),
                                         _p(scale_b, torch.float32), _p(shift_b, torch.float32), M, C_, act_a, act, _p(out),
                                         _stream()), "pai_bn2_add_act")


def bn_bwd_partial_rows(M) -> int:
    return L.load().pai_bn_bwd_partial_rows(M)


def bn_bwd_reduce(dtype, g1, act1, g2, act2, a, z, M, C_, mean, rstd, du, partials, sums, dgamma, dbeta):
    # reads g1 (+ g2), the stored activation (+ z when it is a separate tensor), writes du
    streams = 2 + (g2 is not None) + (a is not None and z is not None and a.data_ptr() != z.data_ptr()) + (du is not None)
    with _TimedBytes("bn_passes", streams * M * C_ * _es(dtype)):
        L.check(L.load().pai_bn_bwd_reduce(code_of(dtype), _p(g1), act1, _p(g2), act2, _p(a), _p(z), M, C_, _p(mean),
                                           _p(rstd), _p(du), _p(partials), _p(sums), _p(dgamma), _p(dbeta),
                                           _stream()), "pai_bn_bwd_reduce")


def bn_bwd_reduce_affine(dtype, g1, act1, g2, act2, z, M, C_, scale, shift, mean, rstd, du, partials, sums, dgamma, dbeta):
    with _TimedBytes("bn_passes", (2 + (g2 is not None) + (du is not None)) * M * C_ * _es(dtype)):
        L.check(L.load().pai_bn_bwd_reduce_affine(code_of(dtype), _p(g1), act1, _p(g2), act2, _p(z), M, C_, _p(scale),
                                                  _p(shift), _p(mean), _p(rstd), _p(du), _p(partials), _p(sums), _p(dgamma),
                                                  _p(dbeta), _stream()), "pai_bn_bwd_reduce_affine")


def bn_bwd_apply(dtype, du, z, M, C_, mean, rstd, gamma, sums, dz):
    with _TimedBytes("bn_passes", 3 * M * C_ * _es(dtype)):      # read du and z, write dz
        L.check(L.load().pai_bn_bwd_apply(code_of(dtype), _p(du), _p(z), M, C_, _p(mean), _p(rstd), _p(gamma),
                                          _p(sums), _p(dz), _stream()), "pai_bn_bwd_apply")


def bn_bwd_apply_affine(dtype, g1, act1, z, M, C_, scale, shift, mean, rstd, gamma, sums, dz):
    """Pass 2 behind ``bn_bwd_reduce_affine(..., du=None, ...)``: du is rebuilt from g1 and z."""
    with _TimedBytes("bn_passes", 3 * M * C_ * _es(dtype)):      # read g1 and z, write dz
        L.check(L.load().pai_bn_bwd_apply_affine(code_of(dtype), _p(g1), act1, _p(z), M, C_, _p(scale), _p(shift), _p(mean),
                                                 _p(rstd), _p(gamma), _p(sums), _p(dz), _stream()), "pai_bn_bwd_apply_affine")


def bn2_bwd_reduce(dtype, d, act_a, za, zb, M, C_, scale_a, shift_a, mean_a, rstd_a, mean_b, rstd_b, part_a, part_b, sums_a,
                   sums_b):
    """Pass 1 of BOTH BatchNorm backwards of a residual block's tail (+ their finalizes): pai_bn2_bwd_reduce."""
    f = torch.float32
    with _TimedBytes("bn_passes", 3 * M * C_ * _es(dtype)):
        L.check(L.load().pai_bn2_bwd_reduce(code_of(dtype), _p(d), int(act_a), _p(za), _p(zb), M, C_, _p(scale_a, f),
                                            _p(shift_a, f), _p(mean_a, f), _p(rstd_a, f), _p(mean_b, f), _p(rstd_b, f),
                                            _p(part_a, f), _p(part_b, f), _p(sums_a, f), _p(sums_b, f), _stream()),
                "pai_bn2_bwd_reduce")


def bn2_bwd_apply(dtype, d, act_a, za, zb, M, C_, scale_a, shift_a, mean_a, rstd_a, gamma_a, sums_a, mean_b, rstd_b, gamma_b,
                  sums_b, dza, dzb):
    f = torch.float32
    with _TimedBytes("bn_passes", 5 * M * C_ * _es(dtype)):
        L.check(L.load().pai_bn2_bwd_apply(code_of(dtype), _p(d), int(act_a), _p(za), _p(zb), M, C_, _p(scale_a, f),
                                           _p(shift_a, f), _p(mean_a, f), _p(rstd_a, f), _p(gamma_a, f), _p(sums_a, f),
                                           _p(mean_b, f), _p(rstd_b, f), _p(gamma_b, f), _p(sums_b, f), _p(dza), _p(dzb),
                                           _stream()), "pai_bn2_bwd_apply")


def act_bwd(dtype, g1, act1, g2, act2, a, numel, du):
    L.check(L.load().pai_act_bwd(code_of(dtype), _p(g1), act1, _p(g2), act2, _p(a), numel, _p(du), _stream()),
            "pai_act_bwd")


# ---- residual U-Net building blocks (models/res_unet.py:199,231,74) -------------------------------
def maxpool2(dtype, x, N, H, W, C_, out, idx=None):
    L.check(L.load().pai_maxpool2(code_of(dtype), _p(x), N, H, W, C_, _p(out), _p(idx, torch.uint8), _stream()),
            "pai_maxpool2")


def maxpool2_bwd(dtype, dout, idx, N, H, W, C_, dx):
    L.check(L.load().pai_maxpool2_bwd(code_of(dtype), _p(dout), _p(idx, torch.uint8), N, H, W, C_, _p(dx), _stream()),
            "pai_maxpool2_bwd")


def upsample2(dtype, x, N, H, W, C_, out):
    L.check(L.load().pai_upsample2(code_of(dtype), _p(x), N, H, W, C_, _p(out), _stream()), "pai_upsample2")


def upsample2_bwd(dtype, dout, N, H, W, C_, dx):
    L.check(L.load().pai_upsample2_bwd(code_of(dtype), _p(dout), N, H, W, C_, _p(dx), _stream()), "pai_upsample2_bwd")


def add_act(dtype, a, b, act, out):
    L.check(L.load().pai_add_act(code_of(dtype), _p(a), _p(b), a.numel(), int(act), _p(out), _stream()), "pai_add_act")


def instnorm_fwd(dtype, x, N, HW, C_, eps, act, y, mean, rstd):
    """y = act(InstanceNorm2d(x)) over an NHWC tensor; mean / rstd [N][C] are kept for the backward pass."""
    L.check(L.load().pai_instnorm_fwd(code_of(dtype), _p(x), N, HW, C_, float(eps), int(act), _p(y), _p(mean, torch.float32),
                                      _p(rstd, torch.float32), _stream()), "pai_instnorm_fwd")


def instnorm_bwd(dtype, g, x, N, HW, C_, act, mean, rstd, dx):
    L.check(L.load().pai_instnorm_bwd(code_of(dtype), _p(g), _p(x), N, HW, C_, int(act), _p(mean, torch.float32),
                                      _p(rstd, torch.float32), _p(dx), _stream()), "pai_instnorm_bwd")


def dropout2d(dtype, x, mask, N, HW, C_, out):
    """out = x * mask[n][c] over an NHWC tensor (nn.Dropout2d forward, and its backward on gradients)."""
    L.check(L.load().pai_dropout2d(code_of(dtype), _p(x), _p(mask, torch.float32), N, HW, C_, _p(out), _stream()),
            "pai_dropout2d")


# ---- TransUNet token ops (models/trans_unet.py:120-180) ------------------------------------------
def layernorm_partial_rows(M) -> int:
    return L.load().pai_layernorm_partial_rows(M)


def layernorm_fwd(dtype, x, res, M, D, gamma, beta, eps, post, P, sum_out, y, mean, rstd):
    L.check(L.load().pai_layernorm_fwd(code_of(dtype), _p(x), _p(res), M, D, _p(gamma, torch.float32),
                                       _p(beta, torch.float32), float(eps), _p(post, torch.float32), int(P),
                                       _p(sum_out), _p(y), _p(mean, torch.float32), _p(rstd, torch.float32), _stream()),
            "pai_layernorm_fwd")


def layernorm_bwd(dtype, dy, xs, M, D, gamma, mean, rstd, dx, dgb=None, partials=None):
    L.check(L.load().pai_layernorm_bwd(code_of(dtype), _p(dy), _p(xs), M, D, _p(gamma, torch.float32),
                                       _p(mean, torch.float32), _p(rstd, torch.float32), _p(dx),
                                       _p(dgb, torch.float32), _p(partials, torch.float32), _stream()),
            "pai_layernorm_bwd")


def gelu(dtype, z, out):
    L.check(L.load().pai_gelu(code_of(dtype), _p(z), z.numel(), _p(out), _stream()), "pai_gelu")


def gelu_bwd(dtype, dy, z, dz):
    L.check(L.load().pai_gelu_bwd(code_of(dtype), _p(dy), _p(z), z.numel(), _p(dz), _stream()), "pai_gelu_bwd")


def mha_fwd(dtype, qkv, S, B, heads, hd, out, probs, mask=None):
    L.check(L.load().pai_mha_fwd(code_of(dtype), _p(qkv), S, B, heads, hd, _p(mask, torch.float32), _p(out),
                                 _p(probs, torch.float32), _stream()), "pai_mha_fwd")


def mha_bwd(dtype, dout, qkv, probs, S, B, heads, hd, dqkv, ds_ws, mask=None):
    L.check(L.load().pai_mha_bwd(code_of(dtype), _p(dout), _p(qkv), _p(probs, torch.float32), S, B, heads, hd,
                                 _p(mask, torch.float32), _p(dqkv), _p(ds_ws, torch.float32), _stream()), "pai_mha_bwd")


def subsample2(dtype, x, N, H, W, C_, out):
    L.check(L.load().pai_subsample2(code_of(dtype), _p(x), N, H, W, C_, _p(out), _stream()), "pai_subsample2")


def subsample2_bwd(dtype, dout, N, H, W, C_, dx):
    L.check(L.load().pai_subsample2_bwd(code_of(dtype), _p(dout), N, H, W, C_, _p(dx), _stream()), "pai_subsample2_bwd")


def bn_stats_rows(M) -> int:
    return L.load().pai_bn_stats_rows(M)


def bn_stats(dtype, z, M, C_, stats):
    L.check(L.load().pai_bn_stats(code_of(dtype), _p(z), M, C_, _p(stats, torch.float32), _stream()), "pai_bn_stats")


def colsum(dtype, x, rows, C_, out):
    """out[C] += column sums of x [rows][C]."""
    L.check(L.load().pai_colsum(code_of(dtype), _p(x), rows, C_, _p(out, torch.float32), _stream()), "pai_colsum")


# ---- attention gate (models/attention_unet.py:88-96) -------------------------------------------
def gate_partial_rows(M) -> int:
    return L.load().pai_gate_partial_rows(M)


def gate_hidden(dtype, ig, sg, M, K, scale_i, shift_i, scale_s, shift_s, w_a, b_a, h, logit, partials):
    L.check(L.load().pai_gate_hidden(code_of(dtype), _p(ig), _p(sg), M, K, _p(scale_i), _p(shift_i), _p(scale_s),
                                     _p(shift_s), _p(w_a, torch.float32), _p(b_a, torch.float32), _p(h),
                                     _p(logit, torch.float32), _p(partials, torch.float32), _stream()),
            "pai_gate_hidden")


def gate_apply(dtype, x, logit, M, C_, scale_a, shift_a, out, att):
    L.check(L.load().pai_gate_apply(code_of(dtype), _p(x), _p(logit, torch.float32), M, C_, _p(scale_a), _p(shift_a),
                                    _p(out), _p(att, torch.float32), _stream()), "pai_gate_apply")


def gate_apply_bwd(dtype, dout, x, att, logit, M, C_, mean_a, rstd_a, dx_skip, dl, partials, relu_out=False):
    L.check(L.load().pai_gate_apply_bwd(code_of(dtype), _p(dout), _p(x), _p(att, torch.float32),
                                        _p(logit, torch.float32), M, C_, _p(mean_a), _p(rstd_a), _p(dx_skip),
                                        _p(dl, torch.float32), _p(partials, torch.float32), int(relu_out), _stream()),
            "pai_gate_apply_bwd")


def gate_hidden_bwd(dtype, dl, logit, h, ig, sg, M, K, mean_a, rstd_a, gamma_a, sums_a, w_a, mean_i, rstd_i,
                    mean_s, rstd_s, dsum, partials_i, partials_s, dw_a, db_a):
    L.check(L.load().pai_gate_hidden_bwd(code_of(dtype), _p(dl, torch.float32), _p(logit, torch.float32), _p(h),
                                         _p(ig), _p(sg), M, K, _p(mean_a), _p(rstd_a), _p(gamma_a), _p(sums_a),
                                         _p(w_a, torch.float32), _p(mean_i), _p(rstd_i), _p(mean_s), _p(rstd_s),
                                         _p(dsum), _p(partials_i, torch.float32), _p(partials_s, torch.float32),
                                         _p(dw_a, torch.float32), _p(db_a, torch.float32), _stream()),
            "pai_gate_hidden_bwd")


def bce_logits(logits, target, loss_scale, loss, grad_scale=0.0, grad=None):
    L.check(L.load().pai_bce_logits(_p(logits, torch.float32), logits.numel(), float(target), float(loss_scale),
                                    _p(loss, torch.float64), float(grad_scale), _p(grad, torch.float32),
                                    _stream()), "pai_bce_logits")


def l1(pred, target, loss_scale, loss, grad_scale=0.0, grad=None):
    L.check(L.load().pai_l1(_p(pred, torch.float32), _p(target, torch.float32), pred.numel(), float(loss_scale),
                            _p(loss, torch.float64), float(grad_scale), _p(grad, torch.float32), _stream()),
            "pai_l1")


def mse(pred, target, loss_scale, loss, grad_scale=0.0, grad=None):
    L.check(L.load().pai_mse(_p(pred, torch.float32), _p(target, torch.float32), pred.numel(), float(loss_scale),
                             _p(loss, torch.float64), float(grad_scale), _p(grad, torch.float32), _stream()),
            "pai_mse")


def scalar_take(acc, out):
    """out = float(acc); acc = 0 (one single-thread launch)."""
    L.check(L.load().pai_scalar_take(_p(acc, torch.float64), _p(out, torch.float32), _stream()), "pai_scalar_take")


def metrics_take(sums, n_images, numel, out3):
    """{sum SSIM, SSE} -> out3 = {mean SSIM, PSNR, RMSE}; sums = 0."""
    L.check(L.load().pai_metrics_take(_p(sums, torch.float64), int(n_images), int(numel), _p(out3, torch.float32),
                                      _stream()), "pai_metrics_take")


def tanh_bwd(dtype, pred, g_a, g_b, dh):
    L.check(L.load().pai_tanh_bwd(code_of(dtype), _p(pred, torch.float32), _p(g_a, torch.float32),
                                  _p(g_b, torch.float32), pred.numel(), _p(dh), _stream()), "pai_tanh_bwd")


def denormalize(x, grad_out, out):
    L.check(L.load().pai_denormalize(_p(x, torch.float32), _p(grad_out, torch.float32), x.numel(),
                                     _p(out, torch.float32), _stream()), "pai_denormalize")


def ssim_sse(pred, target, NC, H, W, denorm, out2=None, per_image=None, full_map=None):
    L.check(L.load().pai_ssim_sse(_p(pred, torch.float32), _p(target, torch.float32), NC, H, W, int(denorm),
                                  _p(out2, torch.float64), _p(per_image, torch.float64),
                                  _p(full_map, torch.float32), _stream()), "pai_ssim_sse")


def ssim_bwd_workspace_floats(NC, H, W) -> int:
    return L.load().pai_ssim_bwd_workspace_floats(NC, H, W)


def ssim_psnr_bwd(pred, target, NC, H, W, denorm, w_ssim, w_psnr, sse, grad, workspace):
    L.check(L.load().pai_ssim_psnr_bwd(_p(pred, torch.float32), _p(target, torch.float32), NC, H, W, int(denorm),
                                       float(w_ssim), float(w_psnr), _p(sse, torch.float64),
                                       _p(grad, torch.float32), _p(workspace, torch.float32), _stream()),
            "pai_ssim_psnr_bwd")


def cast(src, dst):
    assert src.numel() == dst.numel()
    L.check(L.load().pai_cast(code_of(src.dtype), _p(src), code_of(dst.dtype), _p(dst), src.numel(), _stream()),
            "pai_cast")


def cast_multi(pairs) -> None:
    """``cast(src, dst)`` for up to 8 (src, dst) pairs of ONE dtype pair in one launch (pai_cast_multi); pairs that do not
    fit its contract (a multiple of 8 elements, 16-byte aligned) go through ``cast`` one by one."""
    pairs = list(pairs)
    ok = len(pairs) > 1 and len(pairs) <= 8 and len({(s.dtype, d.dtype) for s, d in pairs}) == 1 and all(
        s.numel() == d.numel() and s.numel() % 8 == 0 and s.data_ptr() % 16 == 0 and d.data_ptr() % 16 == 0 and s.is_cuda
        and d.is_cuda and s.is_contiguous() and d.is_contiguous() for s, d in pairs)
    if not ok:
        for s, d in pairs:
            cast(s, d)
        return
    n = len(pairs)
    srcs = (C.c_void_p * n)(*[s.data_ptr() for s, _ in pairs])
    dsts = (C.c_void_p * n)(*[d.data_ptr() for _, d in pairs])
    numels = (C.c_int64 * n)(*[s.numel() for s, _ in pairs])
    L.check(L.load().pai_cast_multi(n, code_of(pairs[0][0].dtype), srcs, code_of(pairs[0][1].dtype), dsts, numels, _stream()),
            "pai_cast_multi")


def reduce_rows(partial, rows, C_, out, accumulate=False):
    L.check(L.load().pai_reduce_rows(_p(partial, torch.float32), rows, C_, _p(out, torch.float32),
                                     int(accumulate), _stream()), "pai_reduce_rows")


def adam(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step):
    with _TimedBytes("adam", 28 * param.numel()):      # read g, p, m, v; write p, m, v
        L.check(L.load().pai_adam(_p(param, torch.float32), _p(grad, torch.float32), _p(exp_avg, torch.float32),
                                  _p(exp_avg_sq, torch.float32), param.numel(), lr, beta1, beta2, eps, step,
                                  _stream()), "pai_adam")


def adam_pack(param, grad, exp_avg, exp_avg_sq, w_off, cout, taps, cin, w_fwd, w_dgrad, lr, beta1, beta2, eps, step):
    """pai_adam_pack: Adam over a range holding one dense conv weight (at element ``w_off``) whose bf16 packs are written by
    the same launch."""
    dt = torch.bfloat16
    packs = 2 * cout * taps * cin * ((w_fwd is not None) + (w_dgrad is not None))
    with _TimedBytes("adam", 28 * param.numel() + packs):
        L.check(L.load().pai_adam_pack(_p(param, torch.float32), _p(grad, torch.float32), _p(exp_avg, torch.float32),
                                       _p(exp_avg_sq, torch.float32), param.numel(), int(w_off), cout, taps, cin,
                                       _p(w_fwd, dt) if w_fwd is not None else None,
                                       _p(w_dgrad, dt) if w_dgrad is not None else None, lr, beta1, beta2, eps, step,
                                       _stream()), "pai_adam_pack")


def adam_dev(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step_dev, coeff_dev):
    """pai_adam_dev: the step count lives in ``step_dev`` (int64 device scalar, advanced by the call)."""
    L.check(L.load().pai_adam_dev(_p(param, torch.float32), _p(grad, torch.float32), _p(exp_avg, torch.float32),
                                  _p(exp_avg_sq, torch.float32), param.numel(), lr, beta1, beta2, eps,
                                  _p(step_dev, torch.int64), _p(coeff_dev, torch.float32), _stream()), "pai_adam_dev")


def adam_multi(params, grads, exp_avgs, exp_avg_sqs, lr, beta1, beta2, eps, step):
    """pai_adam over lists of separately allocated fp32 tensors (a few launches for the whole list)."""
    n = len(params)
    for t in (*params, *grads, *exp_avgs, *exp_avg_sqs):
        if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
            raise PaiError("adam_multi needs contiguous fp32 HIP tensors")
    arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])      # noqa: E731
    numels = (C.c_int64 * n)(*[p.numel() for p in params])
    with _TimedBytes("adam", 28 * sum(p.numel() for p in params)):
        L.check(L.load().pai_adam_multi(n, arr(params), arr(grads), arr(exp_avgs), arr(exp_avg_sqs), numels, lr, beta1,
                                        beta2, eps, step, _stream()), "pai_adam_multi")


def adam_multi_dev(params, grads, exp_avgs, exp_avg_sqs, lr, beta1, beta2, eps, step_dev, coeff_dev):
    """adam_multi with the step count on the device (pai_adam_multi_dev; hipGraph capture)."""
    n = len(params)
    for t in (*params, *grads, *exp_avgs, *exp_avg_sqs):
        if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
            raise PaiError("adam_multi_dev needs contiguous fp32 HIP tensors")
    arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])      # noqa: E731
    numels = (C.c_int64 * n)(*[p.numel() for p in params])
    L.check(L.load().pai_adam_multi_dev(n, arr(params), arr(grads), arr(exp_avgs), arr(exp_avg_sqs), numels, lr, beta1,
                                        beta2, eps, _p(step_dev, torch.int64), _p(coeff_dev, torch.float32), _stream()),
            "pai_adam_multi_dev")


# ---- streams, events, launch plans (include/pai_hip.h: "Launch plans") --------------------------------------------
def _raw(stream) -> int:
    return stream.cuda_stream if hasattr(stream, "cuda_stream") else int(stream)


def stream_wait(waiting, signalling) -> None:
    """``waiting.wait_stream(signalling)`` through the C ABI (pai_stream_wait), so that the edge is part of a plan being
    recorded.  Takes torch streams or raw handles."""
    L.check(L.load().pai_stream_wait(_raw(waiting), _raw(signalling)), "pai_stream_wait")


def stream_wait_last(waiting, signalling) -> None:
    """``stream_wait`` for an edge whose source is the LAST pai launch on ``signalling`` (nothing else was enqueued there
    since): in a recorded plan the source launch carries the event itself (pai_stream_wait_last)."""
    L.check(L.load().pai_stream_wait_last(_raw(waiting), _raw(signalling)), "pai_stream_wait_last")


class Event:
    """A caller-owned event of the C ABI (pai_event_*): ``record(stream)`` now, ``wait(stream)`` later."""

    def __init__(self):
        self._h = C.c_void_p()
        L.check(L.load().pai_event_create(C.byref(self._h)), "pai_event_create")

    def record(self, stream=None):
        L.check(L.load().pai_event_record(self._h, _raw(stream) if stream is not None else _stream()), "pai_event_record")

    def wait(self, stream=None):
        L.check(L.load().pai_stream_wait_event(_raw(stream) if stream is not None else _stream(), self._h),
                "pai_stream_wait_event")

    # never destroyed: a recorded plan may hold the handle (see the header); one event per side stream per process


def zero_multi(tensors) -> None:
    """t.zero_() for a list of contiguous fp32 HIP tensors in one launch per 96 (pai_zero_multi)."""
    n = len(tensors)
    if n == 0:
        return
    for t in tensors:
        if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
            raise PaiError("zero_multi needs contiguous fp32 HIP tensors")
    ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in tensors])
    numels = (C.c_int64 * n)(*[t.numel() for t in tensors])
    L.check(L.load().pai_zero_multi(n, ptrs, numels, _stream()), "pai_zero_multi")


def lerp_multi(segments, weight: float) -> None:
    """dst -= weight * (dst - src) over (dst_ptr, src_ptr, numel) fp32 segments (pai_lerp_multi): the EMA update."""
    n = len(segments)
    if n == 0:
        return
    dsts = (C.c_void_p * n)(*[int(d) for d, _, _ in segments])
    srcs = (C.c_void_p * n)(*[int(s_) for _, s_, _ in segments])
    numels = (C.c_int64 * n)(*[int(k) for _, _, k in segments])
    with _TimedBytes("ema", 12 * sum(int(k) for _, _, k in segments)):
        L.check(L.load().pai_lerp_multi(n, dsts, srcs, numels, float(weight), _stream()), "pai_lerp_multi")


def scale_(t, factor: float) -> None:
    """t *= factor for a contiguous fp32 HIP tensor (pai_scale)."""
    if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
        raise PaiError("scale_ needs a contiguous fp32 HIP tensor")
    L.check(L.load().pai_scale(t.data_ptr(), t.numel(), float(factor), _stream()), "pai_scale")


def filter_to_dense(w_oihw, Cout, cig, taps, groups, dense) -> None:
    """nn.Conv2d filter [Cout][Cin / groups][kh][kw] -> dense tap-major fp32 [Cout][kh * kw][Cin] (pai_filter_to_dense)."""
    L.check(L.load().pai_filter_to_dense(_p(w_oihw, torch.float32), Cout, cig, taps, groups, _p(dense, torch.float32), _stream()),
            "pai_filter_to_dense")


def filter_grad_from_dense(dense_dw, Cout, cig, taps, groups, dw_oihw) -> None:
    L.check(L.load().pai_filter_grad_from_dense(_p(dense_dw, torch.float32), Cout, cig, taps, groups,
                                                _p(dw_oihw, torch.float32), _stream()), "pai_filter_grad_from_dense")


def swap_mid(src, A, B, Cc, D, dst) -> None:
    """dst[a][c][b][:] = src[a][b][c][:] (pai_swap_mid); src / dst contiguous, same dtype (bf16 or fp32)."""
    if src.dtype != dst.dtype or src.numel() != A * B * Cc * D or dst.numel() != src.numel() \
            or not src.is_contiguous() or not dst.is_contiguous():
        raise PaiError("swap_mid: src / dst must be contiguous tensors of A * B * C * D elements of one dtype")
    L.check(L.load().pai_swap_mid(src.element_size(), _p(src), A, B, Cc, D, _p(dst), _stream()), "pai_swap_mid")


class ZeroList:
    """``zero_multi`` of a FIXED tensor list with the pointer tables built once (a gradient arena clears the same ~45
    segments in front of every backward pass)."""

    def __init__(self, tensors):
        for t in tensors:
            if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
                raise PaiError("ZeroList needs contiguous fp32 HIP tensors")
        self.keep = list(tensors)
        self.n = len(self.keep)
        self.ptrs = (C.c_void_p * max(self.n, 1))(*[t.data_ptr() for t in self.keep])
        self.numels = (C.c_int64 * max(self.n, 1))(*[t.numel() for t in self.keep])

    def __call__(self):
        if self.n:
            L.check(L.load().pai_zero_multi(self.n, self.ptrs, self.numels, _stream()), "pai_zero_multi")


class Plan:
    """A recorded launch sequence of libpai_hip.so (pai_plan_*).  ``with plan.recording(): ...`` executes the body
    normally while every launch the library makes (from any thread) is appended; ``run(step_delta)`` re-issues them."""

    def __init__(self):
        self._h = C.c_void_p()
        L.check(L.load().pai_plan_create(C.byref(self._h)), "pai_plan_create")
        self.recorded = False

    def begin(self):
        L.check(L.load().pai_plan_begin(self._h), "pai_plan_begin")

    def end(self):
        L.check(L.load().pai_plan_end(self._h), "pai_plan_end")
        self.recorded = True

    def recording(self):
        plan = self

        class _Ctx:
            def __enter__(self_):
                plan.begin()
                return plan

            def __exit__(self_, *exc):
                plan.end()
                return False
        return _Ctx()

    def run(self, step_delta: int = 0):
        rc = self._run(self._h, step_delta)
        if rc:
            L.check(rc, "pai_plan_run")

    @property
    def _run(self):
        return L.load().pai_plan_run

    def info(self) -> dict:
        a, b, c, r = C.c_int(), C.c_int(), C.c_int(), C.c_int64()
        L.check(L.load().pai_plan_info(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(r)), "pai_plan_info")
        return {"launches": a.value, "waits": b.value, "streams": c.value, "runs": r.value}

    def destroy(self):
        if self._h:
            L.check(L.load().pai_plan_destroy(self._h), "pai_plan_destroy")
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:       # interpreter shutdown / still recording
            pass


class Comm:
    """RCCL communicator behind the C ABI (pai_comm_* / pai_allreduce): in-place SUM all-reduce of device tensors on the
    current stream.  ``unique_id()`` on rank 0, ship the 128 bytes to the other ranks, then ``Comm(id, rank, world)``
    on every rank (the current device is the rank's GPU)."""

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        L.check(L.load().pai_comm_unique_id(buf), "pai_comm_unique_id")
        return buf.raw

    def __init__(self, uid: bytes, rank: int, world: int):
        if len(uid) != 128:
            raise PaiError("Comm: the unique id is 128 bytes")
        self._h = C.c_void_p()
        L.check(L.load().pai_comm_init(C.create_string_buffer(uid, 128), rank, world, C.byref(self._h)), "pai_comm_init")
        self.rank, self.world = rank, world

    def all_reduce(self, t: torch.Tensor):
        if not t.is_cuda or not t.is_contiguous():
            raise PaiError("Comm.all_reduce needs a contiguous HIP tensor")
        L.check(L.load().pai_allreduce(self._h, t.data_ptr(), t.numel(), code_of(t.dtype), _stream()), "pai_allreduce")

    def destroy(self):
        if self._h:
            L.check(L.load().pai_comm_destroy(self._h), "pai_comm_destroy")
            self._h = C.c_void_p()

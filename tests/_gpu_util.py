"""Helpers for the -m gpu parity tests: NCHW <-> NHWC marshalling and C-ABI conv calls."""
import numpy as np
import torch


def dev():
    return torch.device("cuda:0")


def nhwc(t: torch.Tensor, dtype) -> torch.Tensor:
    """NCHW cpu fp32 -> flat NHWC device tensor of the storage dtype."""
    return t.permute(0, 2, 3, 1).contiguous().to(dev()).to(dtype).reshape(-1)


def from_nhwc(flat: torch.Tensor, n, h, w, c) -> torch.Tensor:
    return flat.float().cpu().view(n, h, w, c).permute(0, 3, 1, 2).contiguous()


def fwd_pack(w: torch.Tensor, transposed: bool) -> torch.Tensor:
    """torch weight -> fp32 [Cout][kh][kw][Cin] on device."""
    p = w.permute(1, 2, 3, 0) if transposed else w.permute(0, 2, 3, 1)
    return p.contiguous().to(dev()).float()


def unpack_fwd(flat: torch.Tensor, cout, cin, transposed: bool) -> torch.Tensor:
    p = flat.float().cpu().view(cout, 4, 4, cin)
    return (p.permute(3, 0, 1, 2) if transposed else p.permute(0, 3, 1, 2)).contiguous()


def rnd(shape, seed, scale=1.0):
    rng = np.random.default_rng(seed)
    return torch.from_numpy((rng.standard_normal(shape) * scale).astype(np.float32))


def q(t, dtype):
    """Round a cpu fp32 tensor through the storage dtype (so the CPU reference sees the same inputs)."""
    return t.to(dtype).float()


def rel_err(got: torch.Tensor, want: torch.Tensor) -> float:
    return float((got.double() - want.double()).norm() / max(float(want.double().norm()), 1e-30))


def max_err(got, want) -> float:
    return float((got.double() - want.double()).abs().max() / max(float(want.double().abs().max()), 1e-30))


# ---- training-state copies between two models (launch-plan / graph tests) -------------------------------------
def sync_training_state(src, dst):
    """dst <- src, in place (a recorded plan holds the addresses): parameters, BatchNorm buffers, Adam moments."""
    with torch.no_grad():
        sd = dst.state_dict()
        for k, v in src.state_dict().items():
            sd[k].copy_(v)
        for os_, od in zip(src._all_optimizers(), dst._all_optimizers()):
            if not hasattr(os_, "_engine"):        # MultiAdam: stock per-parameter state, same parameter order
                for ps, pd in zip(os_.param_groups[0]["params"], od.param_groups[0]["params"]):
                    for key in ("exp_avg", "exp_avg_sq"):
                        if key in os_.state.get(ps, ()) and key in od.state.get(pd, ()):
                            od.state[pd][key].copy_(os_.state[ps][key])
                continue
            a_s, a_d = os_._engine.arena(), od._engine.arena()
            if getattr(a_s, "mflat", None) is not None and getattr(a_d, "mflat", None) is not None:
                a_d.mflat.copy_(a_s.mflat)
                a_d.vflat.copy_(a_s.vflat)
        # the bf16 filter packs follow the master weights: the copy above went behind the engines' backs
        for mod in dst.modules():
            eng = getattr(mod, "engine", None) if hasattr(type(mod), "engine") else None
            if eng is not None:
                repack(eng)


def repack(eng):
    """Rewrite the CURRENT packs of every layer from the (just overwritten) master weights, in place."""
    from thesis_pai_reconstruction_amd import ops
    for pk in eng.all_packs():
        if pk.dtype is None or pk.wf is None:
            continue
        w, cout, taps, cin, wf_out, wd_out = pk._prepare(pk.dtype)
        if wf_out is not None or wd_out is not None:
            ops.pack_weights(pk.dtype, w, cout, taps, cin, wf_out, wd_out)
        pk._mark(pk.dtype)

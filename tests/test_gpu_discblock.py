"""GPU: DiscriminatorBlock used on its own -- Conv2d(k4, s2, p1) -> (InstanceNorm2d | Identity) -> LeakyReLU(0.2), reference
models/wrapper.py:176-209 -- against the same stock PyTorch modules on the CPU: forward, input gradient, weight and bias
gradients.  ``norm=True`` (nn.InstanceNorm2d, affine=False) is a branch the reference's own Discriminator never takes
(SURVEY Q4); the class offers it, so it is built (pai_instnorm_fwd / pai_instnorm_bwd) and held to the same bar.
fp32: 1e-4 relative (the parity mode); bf16: the storage rounding."""
import pytest
import torch
import torch.nn as nn

from _gpu_util import dev, rel_err

pytestmark = pytest.mark.gpu


def _reference(cin, cout, norm):
    return nn.Sequential(nn.Conv2d(cin, cout, kernel_size=4, stride=2, padding=1),
                         nn.InstanceNorm2d(cout) if norm else nn.Identity(), nn.LeakyReLU(0.2))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("norm", [False, True], ids=["identity", "instance_norm"])
@pytest.mark.parametrize("shape", [(3, 2, 64, 32, 32), (2, 64, 128, 32, 64), (2, 128, 256, 16, 16), (1, 8, 24, 8, 8)], ids=str)
def test_discriminator_block_alone(pai, shape, norm, dtype):
    N, cin, cout, H, W = shape
    g = torch.Generator().manual_seed(3)
    x = torch.randn(N, cin, H, W, generator=g)
    gy = torch.randn(N, cout, H // 2, W // 2, generator=g)
    ref = _reference(cin, cout, norm)
    with torch.no_grad():
        ref[0].weight.copy_(torch.randn(ref[0].weight.shape, generator=g) * 0.05)
        ref[0].bias.copy_(torch.randn(cout, generator=g) * 0.1)
    if dtype == torch.bfloat16:       # compare against the reference on what the bf16 path can represent
        x, gy = x.bfloat16().float(), gy.bfloat16().float()
    xr = x.clone().requires_grad_(True)
    (ref(xr) * gy).sum().backward()
    want = ref(xr).detach()

    blk = pai.DiscriminatorBlock(cin, cout, norm=norm)
    blk.load_state_dict({"block.0.weight": ref[0].weight.detach(), "block.0.bias": ref[0].bias.detach()})      # = the reference module's keys
    blk.to(dev())
    blk.compute_dtype = dtype
    xd = x.to(dev()).requires_grad_(True)
    got = blk(xd)
    assert got.shape == want.shape and got.dtype == torch.float32
    (got * gy.to(dev())).sum().backward()
    tol = 1e-4 if dtype == torch.float32 else 4e-2      # bf16: filter, activations and gradients are stored in bf16
    assert rel_err(got.detach().cpu(), want) < tol
    assert rel_err(xd.grad.cpu(), xr.grad) < tol
    assert rel_err(blk.block[0].weight.grad.cpu(), ref[0].weight.grad) < tol
    bias_tol = tol if not norm else None       # behind an InstanceNorm the bias gradient is rounding noise around zero
    if bias_tol is not None:
        assert rel_err(blk.block[0].bias.grad.cpu(), ref[0].bias.grad) < bias_tol
    else:
        noise = 1e-3 if dtype == torch.float32 else 2e-2       # (sum of dz values each rounded to the storage type)
        assert float(blk.block[0].bias.grad.abs().max()) <= noise * float(gy.abs().sum(dim=(0, 2, 3)).max())


def test_instance_norm_kernels_against_torch(pai):
    """pai_instnorm_fwd / _bwd alone, fp32, ragged channel count (24) and pixel count (7 x 9), all three activations."""
    from thesis_pai_reconstruction_amd import nnops, ops
    import torch.nn.functional as F
    N, C, H, W = 3, 24, 7, 9
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, C, H, W, generator=g) * 2 + 0.5
    gy = torch.randn(N, C, H, W, generator=g)
    for act, fn in ((ops.ACT_NONE, lambda t: t), (ops.ACT_RELU, F.relu), (ops.ACT_LRELU, lambda t: F.leaky_relu(t, 0.2))):
        xr = x.clone().requires_grad_(True)
        want = fn(F.instance_norm(xr, eps=1e-5))
        (want * gy).sum().backward()
        xd = x.permute(0, 2, 3, 1).contiguous().to(dev()).requires_grad_(True)
        got = nnops.InstanceNormAct.apply(xd, 1e-5, act)
        (got * gy.permute(0, 2, 3, 1).contiguous().to(dev())).sum().backward()
        assert rel_err(got.detach().cpu().permute(0, 3, 1, 2), want.detach()) < 1e-5
        assert rel_err(xd.grad.cpu().permute(0, 3, 1, 2), xr.grad) < 1e-4

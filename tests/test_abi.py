"""CPU: the C-ABI library loads and exports exactly what include/pai_hip.h declares, and the ctypes
table in lib.py covers the same set (no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "pai_hip.h")


def header_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(pai_[a-z0-9_]+)\s*\(", src))


def test_header_lib_and_bindings_agree(pai):
    decl = header_symbols()
    assert len(decl) >= 25
    table = set(pai.lib.SIGNATURES)
    assert decl == table, (sorted(decl - table), sorted(table - decl))
    lib = pai.lib.load()
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in pai_hip.h but not exported by libpai_hip.so"
    assert lib.pai_version() >= 100


def test_descriptor_layout_matches_header(pai):
    # struct pai_conv_desc: 17 int32 fields, no padding
    assert ctypes.sizeof(pai.lib.ConvDesc) == 17 * 4
    src = open(HEADER).read()
    body = src[src.index("typedef struct pai_conv_desc {"):src.index("} pai_conv_desc;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in re.findall(r"int32_t\s+([^;]+);", body):
        for n in decl.split(","):
            names.append(re.sub(r"\[.*\]", "", n).strip())
    assert names == [f[0] for f in pai.lib.ConvDesc._fields_]


def test_host_side_queries_work_without_gpu(pai):
    """Shape / kernel-selection queries are pure host code and must work on a CPU-only box."""
    from thesis_pai_reconstruction_amd import ops
    import torch
    d = ops.make_desc(torch.bfloat16, 0, 64, 128, 128, 64, 0, 128, 2)
    assert ops.conv_out_hw(d) == (64, 64)
    assert ops.conv_kernel_id(d, 0) == 2 and ops.conv_kernel_id(d, 1) in (2, 3) and ops.conv_kernel_id(d, 2) == 2
    # one partial row per output tile: 128-row GEMM tiles or 256-pixel (16x16) patch tiles
    assert ops.conv_fwd_stats_rows(d) in (64 * 64 * 64 // 128, 64 * 64 * 64 // 256)
    assert ops.conv_fwd_stats_rows(d) <= ops.conv_fwd_stats_rows_max(d)
    d32 = ops.make_desc(torch.float32, 0, 64, 128, 128, 64, 0, 128, 2)
    assert ops.conv_kernel_id(d32, 0) == 0            # fp32 storage -> exact-fp32 vector-ALU kernel
    dt = ops.make_desc(torch.bfloat16, 1, 4, 128, 128, 64, 64, 1, 2)
    assert ops.conv_out_hw(dt) == (256, 256) and ops.conv_kernel_id(dt, 0) == 1   # Cout = 1 -> row-dot
    assert ops.conv_flops(d) == 2 * 64 * 64 * 64 * 16 * 64 * 128
    bad = ops.make_desc(torch.float32, 0, 1, 7, 8, 1, 0, 64, 2)
    with pytest.raises(ops.PaiError, match="even"):
        ops.conv_out_hw(bad)


def test_product_path_refuses_cpu_tensors(pai):
    """No CPU fallback: every product entry point raises on host tensors."""
    import torch
    from thesis_pai_reconstruction_amd import functional as PF
    m = pai.Pix2Pix(1, 1, (1, 2), 0.0, "gan")
    x = torch.zeros(1, 1, 32, 32)
    with pytest.raises(pai.PaiError):
        m.unet(x)
    with pytest.raises(pai.PaiError):
        m.discriminator(x, x)
    with pytest.raises(pai.PaiError):
        PF.ssim(x, x)
    with pytest.raises(pai.PaiError):
        PF.l1_loss(x.requires_grad_(True), x)


def test_handle_api_host_side(pai):
    """pai_create / pai_bind / pai_destroy and the per-handle buffer registration are pure host bookkeeping: two handles
    of one device keep their own workspace / scratch, the first one created is active, destroying the active one
    leaves the device without buffers (split-K queries then answer "un-split")."""
    lib = pai.lib.load()
    h1, h2 = ctypes.c_void_p(), ctypes.c_void_p()
    assert lib.pai_create(0, ctypes.byref(h1)) == 0 and lib.pai_create(0, ctypes.byref(h2)) == 0
    assert h1.value and h2.value and h1.value != h2.value
    assert lib.pai_create(-1, ctypes.byref(h1)) != 0 and b"out of range" in lib.pai_last_error()
    fake = ctypes.c_void_p(0x10000)       # never dereferenced on the host
    assert lib.pai_handle_set_workspace(h1, fake, 1 << 20) == 0
    assert lib.pai_handle_set_scratch(h2, fake, 1 << 10) == 0
    assert lib.pai_bind(h2) == 0 and lib.pai_bind(h1) == 0
    assert lib.pai_bind(None) != 0
    assert lib.pai_destroy(h2) == 0 and lib.pai_destroy(h1) == 0
    assert lib.pai_destroy(None) != 0


def test_kernel_dispatch_table_without_gpu(pai):
    """Which kernel a layer runs is host logic: the family of every bf16 case of tests/test_gpu_conv.py and the kernel
    names of the BASELINE configs[1] layers (what bench.py keys its roofline on) are pinned here, without a GPU and
    without a split-K workspace (the un-split names)."""
    import importlib.util
    import torch
    from thesis_pai_reconstruction_amd import ops
    spec = importlib.util.spec_from_file_location("_gpu_conv_cases", os.path.join(ROOT, "tests", "test_gpu_conv.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for name, tr, s, N, H, W, C1, C2, Cout, r1, r2 in mod.CASES:
        d = ops.make_desc(torch.bfloat16, tr, N, H, W, C1, C2, Cout, s, r1, r2, ops.ACT_LRELU)
        assert tuple(ops.conv_kernel_id(d, op) for op in (0, 1, 2)) == mod.BF16_FAMILY[name], name
    big = "gg_fwd_patch_k<256, 128, true>"
    cfg2 = {   # (transposed, N, H, C1, C2, Cout) -> (forward, input gradient, weight gradient)
        "encoders[2]": ((0, 64, 64, 128, 0, 256), (big, big, "gg_wgrad_patch3_k<128, 64, 16>")),
        "decoders[4]": ((1, 64, 16, 512, 512, 256), (big, big, "gg_wgrad_patch3_k<128, 64, 16>")),
        "decoders[5]": ((1, 64, 32, 256, 256, 128), (big, big, "gg_wgrad_patch3_k<128, 64, 16>")),
        "decoders[6]": ((1, 64, 64, 128, 128, 64), ("gg_fwd_patch1_k<256, 64, false>", big, "gg_wgrad_patch3_k<64, 128, 16>")),
        "D block 3": ((0, 128, 32, 256, 0, 512), (big, big, "gg_wgrad_patch3_k<128, 64, 16>")),
    }
    for layer, ((tr, N, H, C1, C2, Cout), want) in cfg2.items():
        d = ops.make_desc(torch.bfloat16, tr, N, H, H, C1, C2, Cout, 2, tr, tr if C2 else 0)
        assert tuple(ops.conv_kernel_name(d, op) for op in (0, 1, 2)) == want, layer

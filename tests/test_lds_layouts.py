"""CPU: the LDS images the implicit-GEMM kernels read MFMA fragments from are bank-conflict-free under the
MI355X banking model (scripts/lds_swizzle_check.py rebuilds every lane's byte address as the kernels do)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fragment_reads_are_conflict_free():
    spec = importlib.util.spec_from_file_location("lds_swizzle_check", os.path.join(ROOT, "scripts", "lds_swizzle_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for fn in (mod.fwd_patch, mod.fwd_weight, mod.fwd_patch32, mod.fwd_weight32, mod.p2_patch, mod.wg_dy, mod.wg_x, mod.wg2_x, mod.wg3_dy, mod.wg3_x, mod.wg3_x8, mod.group_wgrad):
        assert fn() == 1, fn.__name__
    # the model itself: an un-swizzled 128-B-row image read with ds_read_b128 is 2-way
    assert mod.ways(lambda lane: (lane & 15) * 128 + ((lane >> 4) << 4), mod.B128_GROUPS, 16) > 1

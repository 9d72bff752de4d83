"""The hand-scheduled weight-gradient K loops (gg_wgrad_patch3_k<.., PIPE = 1>) read their MFMA fragments from LDS with inline
asm and wait for them with explicit, partly counted s_waitcnt lgkmcnt -- hipcc knows neither.  Two properties of the
GENERATED code carry their correctness and are decided by register allocation, not by the source:
  * nothing reads (or overwrites) a register a ds_read is still to deliver into (a copy hipcc places behind a read, in front
    of its wait, moves garbage: this happened twice while the loop was written);
  * nothing inside the loops goes through scratch memory (a reload is a vmcnt wait that also waits for the LDS-DMA in flight).
scripts/isa_async_check.py walks the ISA of the loops; this test cross-compiles the translation unit (no GPU needed) and
holds the three pipelined kernels to both.  reference: none (implementation property of the HIP path)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def test_pipelined_weight_gradient_loops_never_touch_a_fragment_in_flight(tmp_path):
    import isa_async_check as chk
    sys.path.insert(0, os.path.join(ROOT, "thesis-pai-reconstruction_amd"))
    import build as B
    src = os.path.join(ROOT, "thesis-pai-reconstruction_amd", "csrc", "gg_wg3.hip")
    flags = [f for f in B.FLAGS + B.EXTRA_FLAGS.get("gg_wg3.hip", []) if f not in ("-Wall",)]
    r = subprocess.run([B._hipcc(), *flags, "-c", src, "-o", "gg_wg3.o", "--save-temps"],
                       cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    asm = [f for f in os.listdir(tmp_path) if f.endswith("gfx950.s")]
    assert asm, os.listdir(tmp_path)
    text = open(os.path.join(tmp_path, asm[0])).read()
    pipelined = [n for n, _ in chk.kernels(text) if "gg_wgrad_patch3_k" in n and re.search(r"ELi1EE", n)]
    assert len(pipelined) == 3, pipelined
    rep = chk.check_text(text, pipelined)
    assert len(rep) >= 3
    for (name, a, b), bad in rep.items():
        assert not bad, (name, a, b, bad[:5])
    for name, lines in chk.kernels(text):
        if name not in pipelined:
            continue
        for a, b in chk.loops_with_mfma(lines):
            body = lines[a:b + 1]
            if sum("v_mfma" in l for l in body) < 128:
                continue                      # the two-step steady-state loop (tails may reload an invariant)
            assert not [l for l in body if l.startswith("scratch_")], (name, a, b)

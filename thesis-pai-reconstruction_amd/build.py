"""Builds csrc/*.hip into libpai_hip.so (gfx950 only) with hipcc.

In-tree on purpose: the .so travels with the repository snapshot to the GPU box
(it is git-ignored, not gpurun-ignored).  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpai_hip.so")
STAMPS = os.path.join(CSRC, ".build_stamps.json")     # object / library -> content hash of what it was built from
SOURCES = ["api.hip", "gg_simt.hip", "gg_mfma.hip", "gg_wg3.hip", "gg_thin.hip", "gg_small.hip", "gg_group.hip", "gg_pw.hip", "bn.hip", "ew_stream.hip", "gate.hip", "resnet.hip", "inorm.hip", "vit.hip", "loss.hip", "ssim.hip", "misc.hip", "comm.hip", "plan.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "plan.h"), os.path.join(CSRC, "gg_tile.h"), os.path.join(HERE, "..", "include", "pai_hip.h")]
# -amdgpu-mfma-vgpr-form: keep MFMA accumulators in the (unified) VGPR file.  Without it hipcc 7.2
# put the wgrad accumulators in AGPRs with a different source/destination register per MFMA and
# copied all 64 of them through VGPRs every K-step (192 -> 138 registers, +1 wave per SIMD).
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-fno-gpu-rdc", "-mllvm", "-amdgpu-mfma-vgpr-form=1"]
# Per-source additions.  -fno-slp-vectorize: the SLP pass pairs neighbouring fp32 operations of the tile epilogues into
# v_pk_fma_f32 / v_pk_add_f32 on adjacent registers; at the 128-register line of gg_fwd_patch_k that allocation spilled
# 118-157 registers around the fused producer-backward store (0-4 without the pass), and packed fp32 instructions are the
# slow form beside MFMAs anyway (CDNA4 guide, cycle constants).
EXTRA_FLAGS = {"gg_mfma.hip": ["-fno-slp-vectorize"]}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _digest(paths, extra=""):
    """Content hash of the inputs of one build step (sources, headers, flags): a fresh checkout with shipped
    objects, or a touched-but-unchanged file, must not decide what gets rebuilt -- file times do not travel."""
    import hashlib
    h = hashlib.sha256(extra.encode())
    for p in paths:
        with open(p, "rb") as f:
            h.update(hashlib.sha256(f.read()).digest())
    return h.hexdigest()


def _load_stamps():
    import json
    try:
        with open(STAMPS) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def _save_stamps(stamps):
    import json
    tmp = STAMPS + ".tmp"
    with open(tmp, "w") as f:
        json.dump(stamps, f, indent=0, sort_keys=True)
    os.replace(tmp, STAMPS)


def build_lib(force: bool = False, verbose: bool = True) -> str:
    hipcc = _hipcc()
    stamps = _load_stamps()
    objs, jobs, new = [], [], {}
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        flags = FLAGS + EXTRA_FLAGS.get(src, [])
        new[os.path.basename(o)] = _digest([s] + HEADERS, " ".join(flags))
        if force or not os.path.exists(o) or stamps.get(os.path.basename(o)) != new[os.path.basename(o)]:
            jobs.append([hipcc, *flags, "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    new["libpai_hip.so"] = _digest([], " ".join(new[os.path.basename(o)] for o in objs))
    if force or jobs or not os.path.exists(LIB) or stamps.get("libpai_hip.so") != new["libpai_hip.so"]:
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-ldl"])
    _save_stamps(new)
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv))

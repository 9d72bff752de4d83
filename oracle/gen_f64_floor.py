"""fp32 noise floor of a reference fixture: how far the reference's OWN fp32 gradients are from the
same step evaluated in fp64 (by the oracle, which tests/test_oracle_golden.py pins to the reference).

    python -m oracle.gen_f64_floor ref_att_gan_full

Writes tests/golden/<name>_f64floor.npz with, per generator parameter, the fingerprint distance
(relative, see oracle/fingerprint.py) between the fixture's step-0 gradient and the fp64 gradient.
BatchNorm over the 16-256 samples of the bottleneck attention gates turns fp32 rounding into ~1e-2
relative noise in the reference itself; a parity test cannot ask for less than that.
TEST INFRASTRUCTURE ONLY.
"""
import os
import sys

import numpy as np
import torch

import oracle
from oracle.fingerprint import fingerprint, fingerprint_close
from oracle.gen_golden import synth_batch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(name):
    z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    seed, n, size = int(z["meta.seed"]), int(z["meta.n"]), int(z["meta.size"])
    mults = tuple(int(v) for v in z["meta.mults"])
    fam = str(z["meta.family"]) if "meta.family" in z.files else "pix2pix"
    if fam.startswith("res"):
        g0 = oracle.make_res_unet_state(1, 1, fam[3:], mults)
    else:
        g0 = (oracle.make_attention_unet_state if fam == "attention" else oracle.make_unet_state)(1, 1, mults)
    g = oracle.init_state_portable(g0, seed, perturb_bn=True)
    d = oracle.init_state_portable(oracle.make_disc_state(1), seed + 1)
    g = type(g)((k, v.double() if v.is_floating_point() else v) for k, v in g.items())
    d = type(d)((k, v.double()) for k, v in d.items())
    x, t = synth_batch(seed + 100, n, size)
    if "pred_full" in z.files:
        # forward fixture (oracle/gen_golden.py run_forward_case): distance of the reference's fp32 prediction from the fp64
        # one, relative to max |pred| -- BatchNorm over the TWO samples of a 1 x 1 bottleneck (the 8-level residual U-Net
        # at N = 2) turns fp32 rounding of nearly equal pairs into percent-level noise in the reference itself
        fwd = oracle.res_unet_forward if fam.startswith("res") else \
            (oracle.attention_unet_forward if fam == "attention" else oracle.unet_forward)
        with torch.no_grad():
            pred = fwd(g, x.double(), training=True)
        want = torch.from_numpy(z["pred_full"]).double()
        floor = float((pred - want).abs().max() / want.abs().max())
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", name + "_f64floor.npz"), **{"floor.pred": np.array(floor)})
        print("wrote", name + "_f64floor", {"floor.pred": floor})
        return
    _, grads = oracle.gan_training_step(g, d, oracle.AdamState(), oracle.AdamState(), x.double(), t.double(),
                                        loss_type=str(z["meta.loss_type"]), return_grads=True)
    rec = {}
    for k, gr in grads["g"].items():
        _, worst = fingerprint_close(fingerprint(gr), z[f"step0.ggrad.{k}"], 1.0)
        rec["floor." + k] = np.array(worst)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", name + "_f64floor.npz"), **rec)
    top = sorted(rec.items(), key=lambda kv: -float(kv[1]))[:8]
    print("wrote", name + "_f64floor", [(k, float(v)) for k, v in top])


if __name__ == "__main__":
    torch.set_num_threads(8)
    main(sys.argv[1])

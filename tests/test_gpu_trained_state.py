"""bf16 parity of BASELINE configs[3] / configs[4] at their real widths on a TRAINED state.

At random initialisation the prediction of these networks is rounding-noise dominated (0.54-0.67 relative L2 between
fp32 and bf16 storage: every block opens with a convolution -> BatchNorm that removes a constant and amplifies the 2^-9
storage rounding), so ``tests/test_gpu_configs.py`` could only compare training trajectories there.  Here:

  1. ``oracle/gen_golden.py --trained`` ran the REAL reference classes (``ResUnetGAN("next", (1,2,4,8,8,8,8,8))`` at
     512 x 512 and ``TransUnetGAN((1,2,2,4,4), patch_size=4)``, reference models/res_unet.py:255-315 and
     models/trans_unet.py:50-117) for 40 GAN steps on one batch of two blob pairs and recorded the whole logged
     trajectory, the eval-mode prediction at the end and its per-image SSIM / RMSE.  The trained state itself (0.1-4 GB)
     is not committed.
  2. The fp32 HIP path re-trains from the same portable initialisation with the same inputs and is held to that
     trajectory: tightly over the first steps (one Adam step is sign-SGD -- after it two fp32 implementations drift apart
     like two runs of the reference on different BLAS builds would), within a band afterwards, and its final eval
     prediction to the reference's recorded one.
  3. From THAT trained state the bf16 storage mode must reproduce the fp32 prediction to <= 3 % relative L2, and rank
     the images by SSIM the way the reference's prediction does.
"""
import os

import numpy as np
import pytest
import torch

import oracle
from oracle.gen_golden import blob_batch
from oracle.metrics_ref import ssim_full

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _load(golden_dir, name):
    path = os.path.join(golden_dir, name + ".npz")
    if not os.path.exists(path):
        pytest.skip(f"{name}.npz not generated (python -m oracle.gen_golden --trained)")
    return np.load(path)


def _build(pai, family, mults, seed):
    if family.startswith("trans"):
        m = pai.TransUnetGAN(in_channels=1, out_channels=1, channel_mults=tuple(mults), patch_size=int(family[5:]), dropout=0.0,
                             loss_type="gan")
        g = oracle.init_trans_state_portable(oracle.make_trans_unet_state(1, 1, tuple(mults), int(family[5:])), seed)
    else:
        m = pai.ResUnetGAN(in_channels=1, out_channels=1, res_type=family[3:], channel_mults=tuple(mults), dropout=0.0,
                           loss_type="gan")
        g = oracle.init_state_portable(oracle.make_res_unet_state(1, 1, family[3:], tuple(mults)), seed, perturb_bn=True)
    m.unet.load_state_dict(g, strict=True)
    m.discriminator.load_state_dict(oracle.init_state_portable(oracle.make_disc_state(1), seed + 1), strict=True)
    return m.to(DEV)


def _rel(a, b):
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize("name", ["ref_resnext_trained_full", "ref_trans4_trained_full"])
def test_bf16_prediction_of_a_trained_state_at_full_width(pai, golden_dir, name):
    z = _load(golden_dir, name)
    seed, n, size, steps, fam = int(z["meta.seed"]), int(z["meta.n"]), int(z["meta.size"]), int(z["meta.steps"]), str(z["meta.family"])
    mults = [int(v) for v in z["meta.mults"]]
    keys = [str(k) for k in z["traj.keys"]]
    want = z["traj.values"]
    x, t = blob_batch(seed + 100, n, size)
    x, t = x.to(DEV), t.to(DEV)
    m = _build(pai, fam, mults, seed)
    m.set_precision("32")
    m.train()
    got = []
    for s in range(steps):
        m.logged = {}
        m.training_step((x, t), s)
        got.append([float(m.logged[k]) for k in keys])
    got = np.array(got)
    assert np.isfinite(got).all()
    # (2) the fp32 HIP path against the reference's own trajectory.  Steps 0-2: the bound of the few-step fixtures.  Later the
    # two runs have taken `steps` sign-like Adam steps apart and the discriminator loss of a GAN oscillates (the reference's
    # own curve jumps 1.1 -> 1.6 -> 1.2 between consecutive steps): what is comparable is the SMOOTH side -- RMSE / PSNR of
    # the prediction -- as a band around the reference's curve; the discriminator loss only by its range.
    drift = {}
    for j, k in enumerate(keys):
        for s in range(steps):
            a, b = got[s, j], want[s, j]
            if s < 3:
                tol = 2e-3 if s == 0 else 0.03
                assert abs(a - b) <= tol * max(abs(b), 1.0 if k.endswith("loss") else 0.05), (name, k, s, a, b)
            elif k in ("train_rmse", "train_psnr"):
                assert abs(a - b) <= 0.12 * abs(b), (name, k, s, a, b)      # measured <= 0.036 of the curve's maximum
            elif k == "d_loss":
                assert 0.0 < a < 2.5 * max(want[:, j].max(), 1.0), (name, k, s, a)
        drift[k] = round(float(np.abs(got[:, j] - want[:, j]).max() / max(np.abs(want[:, j]).max(), 1e-9)), 4)
    print(f"{name}: largest trajectory drift per key (relative to the curve's maximum)", drift)
    # both curves learnt the batch
    i_rmse = keys.index("train_rmse")
    assert got[-1, i_rmse] < 0.8 * got[0, i_rmse] and want[-1, i_rmse] < 0.8 * want[0, i_rmse]
    m.eval()
    with torch.no_grad():
        p32 = m.unet(x).float().cpu()
    ref_pred = torch.from_numpy(z["val.pred_full"])
    r = _rel(p32, ref_pred)
    print(f"{name}: fp32 HIP eval prediction vs the reference's after {steps} steps: relative L2 {r:.4f}")
    assert r < 0.15, r            # two diverged-but-equivalent training runs (measured 0.046 / 0.027): the SAME images, not the same bits
    # (3) bf16 storage from the trained fp32 state
    m.set_precision("bf16-mixed")
    with torch.no_grad():
        p16 = m.unet(x).float().cpu()
    rb = _rel(p16, p32)
    print(f"{name}: bf16 vs fp32 prediction on the trained state: relative L2 {rb:.4f} (random init: 0.54-0.67)")
    assert rb <= 0.03, rb         # measured 0.0095 (configs[3]) / 0.0017 (configs[4])
    tt = t.cpu()
    s16 = ssim_full((p16 + 1) / 2, (tt + 1) / 2)[0].numpy()
    s32 = ssim_full((p32 + 1) / 2, (tt + 1) / 2)[0].numpy()
    sref = z["val.ssim_per_image"]
    # per-image SSIM of the two storage modes: the images sit at SSIM ~0.3 after 40 steps, where the statistic is steep in the
    # small-amplitude detail bf16 rounds away: measured 0.003-0.013 from run to run (the fp32 path's few atomics make the
    # trained state itself vary in the last bits); the 5e-3 bound of the fixtures belongs to SSIM ~0.01 predictions
    assert np.abs(s16 - s32).max() <= 2.5e-2, (s16, s32)
    # per-image SSIM ordering of the reference's prediction wherever it separates the images by more than the bf16 bound
    for i in range(n):
        for j in range(n):
            if sref[i] - sref[j] > 0.05:
                assert s16[i] > s16[j] and s32[i] > s32[j], (i, j, sref, s32, s16)

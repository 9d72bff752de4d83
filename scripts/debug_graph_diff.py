"""Where a graphed step starts to differ from the eager one: per step, max |difference| of every logged value and of the
parameters (tiny GAN, bf16).  python scripts/debug_graph_diff.py   (GPU box)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pai_bootstrap  # noqa: E402
import oracle  # noqa: E402
from oracle.gen_golden import synth_batch  # noqa: E402

pai = pai_bootstrap.load()
from thesis_pai_reconstruction_amd.graph import GraphedStep  # noqa: E402

DEV = "cuda:0"


def build(mults, seed, dtype):
    m = pai.Pix2Pix(1, 1, tuple(mults), 0.0, "gan")
    m.unet.load_state_dict(oracle.init_state_portable(oracle.make_unet_state(1, 1, tuple(mults)), seed, perturb_bn=True))
    m.discriminator.load_state_dict(oracle.init_state_portable(oracle.make_disc_state(1), seed + 1))
    m.to(DEV)
    m.set_precision("32" if dtype == torch.float32 else "bf16-mixed")
    m.train()
    return m


def main():
    dtype = torch.bfloat16
    mults, n, size, steps = (1, 2, 4, 8), 4, 64, 7
    batches = [tuple(t.to(DEV) for t in synth_batch(100 + s, n, size)) for s in range(steps)]
    eager, graphed = build(mults, 3, dtype), build(mults, 3, dtype)
    mode = sys.argv[1] if len(sys.argv) > 1 else "graph"
    if mode == "graph":
        gs = GraphedStep(graphed, warmup=2)
    else:
        os.environ["PAI_NO_STREAM_ADAM"] = "1"          # eager without the streamed update
        gs = lambda b, s: graphed.training_step(b, s)
    for s, b in enumerate(batches):
        eager.logged, graphed.logged = {}, {}
        if mode != "graph":
            os.environ["PAI_NO_STREAM_ADAM"] = "0"
        eager.training_step(b, s)
        if mode != "graph":
            os.environ["PAI_NO_STREAM_ADAM"] = "1"
        gs(b, s)
        torch.cuda.synchronize()
        dl = {k: abs(float(v) - float(graphed.logged[k])) for k, v in eager.logged.items()}
        worst = (0.0, "")
        for (k, p), (_, q) in zip(eager.state_dict().items(), graphed.state_dict().items()):
            if p.dtype.is_floating_point:
                d = float((p.float() - q.float()).abs().max())
                if d > worst[0]:
                    worst = (d, k)
        print(s, {k: f"{v:.3g}" for k, v in dl.items()}, "params", f"{worst[0]:.3g}", worst[1], flush=True)


main()

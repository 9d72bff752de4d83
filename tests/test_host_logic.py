"""CPU: host-side logic that does not touch the kernels -- parameter layout / gradient arena,
Lightning-compatible hooks, trainer loop, CSV logger, best-checkpoint selection, state-dict
compatibility with the reference's key names and shapes."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

import oracle


def test_state_dict_keys_and_shapes_match_reference(pai):
    m = pai.Pix2Pix(1, 1, (1, 2, 4, 8, 8, 8, 8, 8), 0.0, "gan")
    g = oracle.make_unet_state(1, 1, (1, 2, 4, 8, 8, 8, 8, 8))
    d = oracle.make_disc_state(1)
    sd = m.state_dict()
    assert len(sd) == 106                      # 106 tensors for Pix2Pix-GAN
    for k, v in g.items():
        assert tuple(sd["unet." + k].shape) == tuple(v.shape), k
    for k, v in d.items():
        assert tuple(sd["discriminator." + k].shape) == tuple(v.shape), k
    assert sum(p.numel() for p in m.unet.parameters()) == 54_413_313
    assert sum(p.numel() for p in m.discriminator.parameters()) == 2_763_712
    assert m.hparams == {"in_channels": 1, "out_channels": 1, "channel_mults": (1, 2, 4, 8, 8, 8, 8, 8),
                         "dropout": 0.0, "loss_type": "gan"}
    assert m.automatic_optimization is False


def test_other_families_state_dicts_and_init(pai):
    """Attention / residual U-Nets: state-dict keys, order and shapes equal the reference's (oracle.make_*_state is
    what tests/test_oracle_golden.py loads the REAL reference's fixtures into), and init_weights leaves conv weights
    ~N(0, 0.02), norm affines at (1, 0) (reference models/utils.py:15-28)."""
    m = pai.AttentionUnetGAN(1, 1, (1, 2, 4, 8), 0.0, "gan")
    want = oracle.make_attention_unet_state(1, 1, (1, 2, 4, 8))
    got = m.unet.state_dict()
    assert [(k, tuple(v.shape)) for k, v in got.items()] == [(k, tuple(v.shape)) for k, v in want.items()]
    for rt in ("18", "50", "next", "v2"):
        r = pai.ResUnetGAN(1, 1, rt, (1, 2, 2), 0.0, "gan")
        want = oracle.make_res_unet_state(1, 1, rt, (1, 2, 2))
        got = r.unet.state_dict()
        assert [(k, tuple(v.shape)) for k, v in got.items()] == [(k, tuple(v.shape)) for k, v in want.items()], rt
    w = m.unet.attention_blocks[0].input_gate[0].weight
    assert abs(float(w.std()) - 0.02) < 4e-3 and abs(float(w.mean())) < 4e-3
    bn = m.unet.attention_blocks[0].input_gate[1]
    assert torch.equal(bn.weight, torch.ones_like(bn.weight)) and torch.equal(bn.bias, torch.zeros_like(bn.bias))


def test_trans_unet_state_dict_and_cpu_refusal(pai):
    """TransUnet: state-dict keys / shapes equal the reference layout (oracle.make_trans_unet_state is what the fixtures of
    the REAL reference load into), the wrapper gets MultiAdam, and the forward refuses CPU tensors (no CPU path exists)."""
    from thesis_pai_reconstruction_amd.optim import MultiAdam
    for mults, patch in (((1, 1, 1, 2, 2), 2), ((1, 1, 1, 1, 1), 4)):
        m = pai.TransUnetGAN(1, 1, mults, patch, 0.0, "gan")
        want = oracle.make_trans_unet_state(1, 1, mults, patch)
        got = m.unet.state_dict()
        assert set(got) == set(want) and all(tuple(got[k].shape) == tuple(want[k].shape) for k in want)
    lin = m.unet.vit_bottleneck.to_patch_embedding[2]
    assert abs(float(lin.weight.std()) - 0.02) < 2e-3                       # init_weights reaches nn.Linear
    ln = m.unet.vit_bottleneck.transformer.layers[3].norm1
    assert torch.equal(ln.weight, torch.ones_like(ln.weight)) and torch.equal(ln.bias, torch.zeros_like(ln.bias))
    opt_g, opt_d = m.configure_optimizers()
    assert isinstance(opt_g, MultiAdam) and not isinstance(opt_d, MultiAdam)
    with pytest.raises(pai.PaiError):
        m.unet(torch.zeros(1, 1, 256, 256))


def test_fwd_pack_layout_and_arena_views(pai):
    from thesis_pai_reconstruction_amd import engine as E
    conv = nn.Conv2d(6, 64, 4, 2, 1)
    convt = nn.ConvTranspose2d(128, 64, 4, 2, 1)
    w0, wt0 = conv.weight.detach().clone(), convt.weight.detach().clone()
    E.to_fwd_pack_(conv)
    E.to_fwd_pack_(convt)
    assert torch.equal(conv.weight, w0) and torch.equal(convt.weight, wt0)       # logical values unchanged
    assert conv.weight.shape == (64, 6, 4, 4) and convt.weight.shape == (128, 64, 4, 4)
    # physical order is [Cout][kh][kw][Cin]
    phys = conv.weight.detach().as_strided((64, 4, 4, 6), (96, 24, 6, 1))
    assert torch.equal(phys, w0.permute(0, 2, 3, 1))
    physt = convt.weight.detach().as_strided((64, 4, 4, 128), (2048, 512, 128, 1))
    assert torch.equal(physt, wt0.permute(1, 2, 3, 0))
    arena = E.GradArena([(conv.weight, conv), (conv.bias, None), (convt.weight, convt)], torch.device("cpu"))
    assert arena.view(conv.weight).shape == conv.weight.shape
    assert arena.view(conv.weight).stride() == conv.weight.stride()
    assert arena.view(convt.weight).stride() == convt.weight.stride()
    arena.seg(convt.weight).copy_(torch.arange(convt.weight.numel(), dtype=torch.float32))
    assert float(arena.view(convt.weight)[5, 3, 2, 1]) == float(((3 * 4 + 2) * 4 + 1) * 128 + 5)
    arena.begin_backward([conv.weight, conv.bias, convt.weight])
    assert float(arena.flat.abs().sum()) == 0
    arena.attach([conv.weight, conv.bias, convt.weight])
    assert conv.weight.grad.data_ptr() == arena.view(conv.weight).data_ptr()
    arena.begin_backward([conv.weight, conv.bias, convt.weight])     # second pass accumulates, no error
    conv.bias.grad = torch.zeros(64)
    with pytest.raises(pai.PaiError):
        arena.begin_backward([conv.weight, conv.bias, convt.weight])
    # state-dict round trip keeps values whatever the physical layout
    sd = {k: v.clone() for k, v in convt.state_dict().items()}
    convt2 = nn.ConvTranspose2d(128, 64, 4, 2, 1)
    E.to_fwd_pack_(convt2)
    convt2.load_state_dict(sd)
    assert torch.equal(convt2.weight, wt0)


class Toy(torch.nn.Module):
    pass


def _toy_module(pai):
    from thesis_pai_reconstruction_amd.lightning import LightningModule

    class ToyGAN(LightningModule):
        """Two optimisers, manual optimisation, same hook sequence as UnetWrapper.training_step."""

        def __init__(self, width=4):
            super().__init__()
            self.automatic_optimization = False
            self.g = nn.Linear(width, width)
            self.d = nn.Linear(width, 1)
            self.save_hyperparameters()
            self.seen = []

        def configure_optimizers(self):
            return (torch.optim.SGD(self.g.parameters(), lr=0.1), torch.optim.SGD(self.d.parameters(), lr=0.1))

        def training_step(self, batch, batch_idx):
            x, t = batch
            opt_g, opt_d = self.optimizers()
            self.toggle_optimizer(opt_d)
            self.seen.append(("d", [p.requires_grad for p in self.g.parameters()],
                              [p.requires_grad for p in self.d.parameters()]))
            dl = self.d(self.g(x)).mean()
            self.d.zero_grad(set_to_none=True)
            self.manual_backward(dl)
            opt_d.step()
            self.untoggle_optimizer(opt_d)
            self.toggle_optimizer(opt_g)
            self.seen.append(("g", [p.requires_grad for p in self.g.parameters()],
                              [p.requires_grad for p in self.d.parameters()]))
            loss = ((self.g(x) - t) ** 2).mean()
            self.log("loss", loss)
            self.g.zero_grad(set_to_none=True)
            self.manual_backward(loss)
            opt_g.step()
            self.untoggle_optimizer(opt_g)

        def validation_step(self, batch, batch_idx):
            x, t = batch
            self.log("val_ssim", -((self.g(x) - t) ** 2).mean())

    return ToyGAN


def test_toggle_optimizer_semantics(pai):
    m = _toy_module(pai)()
    x, t = torch.randn(8, 4), torch.randn(8, 4)
    m.training_step((x, t), 0)
    assert m.seen[0] == ("d", [False, False], [True, True])
    assert m.seen[1] == ("g", [True, True], [False, False])
    assert all(p.requires_grad for p in m.parameters())            # restored afterwards
    assert isinstance(m.optimizers(), list) and len(m.optimizers()) == 2
    assert m.hparams == {"width": 4}


def test_trainer_fit_logs_and_keeps_best_checkpoint(pai, tmp_path):
    from thesis_pai_reconstruction_amd.lightning import CSVLogger, ModelCheckpoint, Trainer
    ToyGAN = _toy_module(pai)
    torch.manual_seed(0)
    m = ToyGAN()
    data = [(torch.randn(8, 4), torch.randn(8, 4)) for _ in range(5)]
    logger = CSVLogger(str(tmp_path / "logs"), name="run")
    ckpt = ModelCheckpoint(save_top_k=1, monitor="val_ssim", mode="max", filename="best")
    tr = Trainer(max_epochs=4, max_steps=-1, log_every_n_steps=2, check_val_every_n_epoch=2, logger=[logger],
                 callbacks=[ckpt], enable_progress_bar=False)
    tr.fit(m, train_dataloaders=data, val_dataloaders=data[:2])
    # Lightning 2.0, manual optimisation: global_step counts optimizer.step() calls -- two per batch for the GAN loss
    # (reference models/wrapper.py:136,160); the CSV "step" column and the logging cadence count batches
    assert tr.batches_seen == 20 and tr.global_step == 40
    csv_path = tmp_path / "logs" / "run" / "version_0" / "metrics.csv"
    rows = open(csv_path).read().strip().splitlines()
    assert rows[0].split(",")[:1] == ["loss"] and "val_ssim" in rows[0] and "step" in rows[0]
    assert len(rows) == 1 + 10 + 2                    # 10 train rows (every 2 steps) + 2 validation rows
    best = tmp_path / "logs" / "run" / "version_0" / "checkpoints" / "best.ckpt"
    assert os.path.exists(best) and ckpt.best_model_path == str(best)
    ck = torch.load(best, weights_only=False)
    assert set(ck) >= {"state_dict", "hyper_parameters", "epoch", "global_step", "optimizer_states", "callbacks"}
    assert len(ck["optimizer_states"]) == 2 and all("state" in o and "param_groups" in o for o in ck["optimizer_states"])
    m2 = ToyGAN.load_from_checkpoint(best)
    m2.freeze()
    assert not any(p.requires_grad for p in m2.parameters()) and not m2.training
    # a second run gets version_1
    assert CSVLogger(str(tmp_path / "logs"), name="run").version == 1
    # max_steps stops mid-epoch
    tr2 = Trainer(max_epochs=100, max_steps=7, logger=None, enable_progress_bar=False)
    tr2.fit(ToyGAN(), train_dataloaders=data)
    assert tr2.global_step == 8 and tr2.batches_seen == 4      # --steps counts optimizer steps: 7 is reached inside batch 4
    # resume: weights, Adam moments / step counts and the counters continue from the checkpoint
    torch.manual_seed(1)
    m3 = ToyGAN()
    tr3 = Trainer(max_epochs=6, enable_progress_bar=False, logger=None)
    tr3.fit(m3, train_dataloaders=data, ckpt_path=best)
    # the best checkpoint was written after epoch index 1 or 3; training continues with the NEXT epoch up to max_epochs = 6
    assert tr3.batches_seen == ck["batches_seen"] + 5 * (6 - ck["epoch"] - 1) and tr3.global_step == 2 * tr3.batches_seen
    # ... and so does the score to beat: a resumed run must not overwrite best.ckpt with a worse first validation
    assert ck["callbacks"]["ModelCheckpoint"]["best_model_score"] == pytest.approx(float(ckpt.best_model_score))
    ckpt4 = ModelCheckpoint(save_top_k=1, monitor="val_ssim", mode="max", filename="best", dirpath=str(tmp_path / "resumed"))
    tr4 = Trainer(max_epochs=ck["epoch"] + 1, enable_progress_bar=False, logger=None, callbacks=[ckpt4])
    tr4.restore(ToyGAN(), best)
    assert ckpt4.best_model_score == pytest.approx(float(ckpt.best_model_score)) and ckpt4.best_model_path == str(best)


def test_precision_strings(pai):
    from thesis_pai_reconstruction_amd.lightning import precision_to_dtype
    assert precision_to_dtype("32") == torch.float32
    assert precision_to_dtype("bf16-mixed") == torch.bfloat16
    with pytest.raises(ValueError):
        precision_to_dtype("8")
    with pytest.raises(ValueError, match="fp16"):
        precision_to_dtype("16-mixed")        # no silent remapping to bf16
    m = pai.Pix2Pix(1, 1, (1, 2), 0.0, "gan")
    m.set_precision("bf16-mixed")
    assert m.unet.compute_dtype == torch.bfloat16 and m.discriminator.compute_dtype == torch.bfloat16


def test_ema_callback_matches_torch_ema_semantics(pai):
    """torch_ema 0.3 update rule with warm-up decay, swap-in / restore around validation
    (reference callbacks/ema.py:24-52)."""
    from thesis_pai_reconstruction_amd.callbacks import EMACallback
    torch.manual_seed(0)
    net = nn.Linear(3, 2)
    cb = EMACallback(decay=0.9)
    cb.on_fit_start(None, net)
    ref = [p.detach().clone() for p in net.parameters()]
    for n in range(1, 6):
        with torch.no_grad():
            for p in net.parameters():
                p.add_(torch.randn_like(p) * 0.1)
        cb.on_train_batch_end(None, net)
        d = min(0.9, (1 + n) / (10 + n))
        ref = [r - (1 - d) * (r - p.detach()) for r, p in zip(ref, net.parameters())]
    for s, r in zip(cb.shadow, ref):
        assert torch.allclose(s, r, atol=1e-6)
    live = [p.detach().clone() for p in net.parameters()]
    cb.on_validation_start(None, net)
    for p, r in zip(net.parameters(), ref):
        assert torch.allclose(p, r, atol=1e-6)
    cb.on_validation_end(None, net)
    for p, l in zip(net.parameters(), live):
        assert torch.equal(p, l)
    assert cb.state_dict()["num_updates"] == 5


def test_rccl_knobs_and_self_description_of_the_gradient_exchange(pai, monkeypatch):
    """the collective's knobs are settable before the communicator exists and the bench line can say
    what exchange it ran (transport, ranks, knobs in force, bucket bytes, wire type)."""
    from thesis_pai_reconstruction_amd import dist as pdist
    for v in pdist.RCCL_KNOBS.values():
        monkeypatch.delenv(v, raising=False)
    assert pdist.configure_rccl() == {"algo": None, "proto": None, "min_channels": None, "max_channels": None}
    got = pdist.configure_rccl(algo="Tree", min_channels=2, max_channels=2)
    assert got["algo"] == "Tree" and os.environ["NCCL_ALGO"] == "Tree" and os.environ["NCCL_MIN_NCHANNELS"] == "2"
    monkeypatch.setenv("PAI_DDP_BUCKET_MB", "8")
    r = pdist.GradReducer()                      # no process group: one rank
    d = r.describe()
    assert d["world"] == 1 and d["rccl_ranks"] == 0 and d["bucket_bytes"] == 8 << 20 and d["wire_dtype"] == "float32"
    assert d["rccl_knobs"]["algo"] == "Tree" and "not RCCL" in d["algorithm"]
    assert pdist.GradReducer(bucket_bytes=4 << 20).describe()["bucket_bytes"] == 4 << 20

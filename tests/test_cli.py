"""CLI surface: flags and defaults of main.py / report.py equal the reference's
(reference main.py:140-230, report.py:237-267); end-to-end train -> checkpoint -> report on the GPU."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (dest, default) recorded from the reference's argparse definitions
REF_TRAIN = {"name": None, "data": None, "val_data": None, "epochs": 200, "steps": -1, "batch_size": 8,
             "val_epochs": 10, "precision": "32", "ema": False, "channel_mults": "1,2,4,8,8,8,8,8",
             "attention_res": "8,4,2", "dropout": 0.0, "loss_type": "gan", "schedule_type": "linear",
             "learn_variance": False, "model": "pix2pix"}
REF_REPORT = {"name": None, "checkpoint": None, "data": None, "batch_size": 2, "model": "pix2pix"}


def _defaults(parser):
    return {a.dest: a.default for a in parser._actions if a.dest != "help"}


def test_train_and_report_flags_match_reference():
    sys.path.insert(0, ROOT)
    import main
    import report
    d = _defaults(main.build_parser())
    for k, v in REF_TRAIN.items():
        assert k in d and d[k] == v, (k, d.get(k), v)
    opts = {s for a in main.build_parser()._actions for s in a.option_strings}
    assert {"-d", "--data", "-vd", "--val-data", "-e", "--epochs", "-s", "--steps", "--batch-size",
            "--val-epochs", "--precision", "--ema", "--no-ema", "--channel-mults", "--attention-res",
            "--dropout", "--loss-type", "--schedule-type", "--learn-variance", "-m", "--model"} <= opts
    r = _defaults(report.build_parser())
    for k, v in REF_REPORT.items():
        assert k in r and r[k] == v, (k, r.get(k), v)
    models = next(a for a in report.build_parser()._actions if a.dest == "model").choices
    assert "identity" in models and "pix2pix" in models


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["pix2pix", "attention_unet", "resnext_unet", "trans_unet"])
def test_train_then_report_roundtrip(tmp_path, model):
    env = dict(os.environ, PYTHONPATH=ROOT)
    # TransUnetGAN fixes image_size = 256 (reference models/trans_unet.py:22) and needs a channel_mults that leaves patches
    shape = (["--synthetic", "8", "--batch-size", "4", "--channel-mults", "1,1,1,1,1", "--image-size", "256"]
             if model == "trans_unet" else
             ["--synthetic", "24", "--batch-size", "8", "--channel-mults", "1,2,2,4", "--image-size", "64"])
    run = subprocess.run([sys.executable, os.path.join(ROOT, "main.py"), "cli_run", *shape, "-e", "2", "--val-epochs", "1",
                          "-m", model], cwd=tmp_path, env=env, capture_output=True, text=True,
                         timeout=600)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    vdir = tmp_path / "logs" / "cli_run" / "version_0"
    assert (vdir / "metrics.csv").exists()
    header = open(vdir / "metrics.csv").readline()
    assert "val_ssim" in header and "val_psnr" in header and "val_rmse" in header
    ckpt = vdir / "checkpoints" / "best.ckpt"
    assert ckpt.exists()
    rep = subprocess.run([sys.executable, os.path.join(ROOT, "report.py"), "cli_rep", "-c", str(ckpt), "-bs", "4",
                          "-m", model],
                         cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert rep.returncode == 0, rep.stdout[-2000:] + rep.stderr[-2000:]
    rdir = tmp_path / "reports" / "cli_rep"
    for f in ("depth_ssim.csv", "stats.txt", "ssim_per_image.csv", "psnr_per_image.csv", "mse_per_image.csv"):
        assert (rdir / f).exists(), f
    assert len(os.listdir(rdir / "outputs")) == 16 and len(os.listdir(rdir / "ssim_images")) == 16
    stats = dict(l.strip().split(": ") for l in open(rdir / "stats.txt"))
    assert set(stats) == {"SSIM", "PSNR", "RMSE", "FLOPs", "Parameter count"}
    assert int(stats["Parameter count"]) > 2_000_000 and 0 < float(stats["SSIM"]) <= 1
    assert len(open(rdir / "depth_ssim.csv").read().strip().splitlines()) == 17

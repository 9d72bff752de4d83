"""Train CLI -- same flags, defaults and output tree as the reference's main.py:139-233
(logs/<name>/version_k/{metrics.csv,checkpoints/best.ckpt}), running on the MI355X hot path.

Added (build-only) flags: --synthetic N (train on N synthetic pairs instead of a YAML list),
--image-size, --num-workers.  Multi-GPU: launch with
``python -m torch.distributed.run --nproc-per-node N main.py ...`` (one process per GPU, RCCL).
"""
import argparse
import pathlib
from argparse import ArgumentParser

import torch

import pai_bootstrap

pai = pai_bootstrap.load()
from thesis_pai_reconstruction_amd import dist as pdist  # noqa: E402
from thesis_pai_reconstruction_amd.dataset import ImageDataModule, SyntheticDataModule  # noqa: E402
from thesis_pai_reconstruction_amd.callbacks import EMACallback  # noqa: E402
from thesis_pai_reconstruction_amd.lightning import CSVLogger, ModelCheckpoint, Trainer  # noqa: E402

RES_TYPES = {"res18_unet": "18", "res50_unet": "50", "resv2_unet": "v2", "resnext_unet": "next"}
HIP_MODELS = ("pix2pix", "attention_unet", "res18_unet", "res50_unet", "resv2_unet", "resnext_unet", "trans_unet")


def main(hparams):
    channel_mults = [int(x) for x in hparams.channel_mults.split(",")]
    if hparams.model == "pix2pix":
        model = pai.Pix2Pix(in_channels=1, out_channels=1, channel_mults=channel_mults,
                            dropout=hparams.dropout, loss_type=hparams.loss_type)
    elif hparams.model == "attention_unet":
        model = pai.AttentionUnetGAN(in_channels=1, out_channels=1, channel_mults=channel_mults,
                                     dropout=hparams.dropout, loss_type=hparams.loss_type)
    elif hparams.model in RES_TYPES:
        model = pai.ResUnetGAN(in_channels=1, out_channels=1, res_type=RES_TYPES[hparams.model],
                               channel_mults=channel_mults, dropout=hparams.dropout, loss_type=hparams.loss_type)
    elif hparams.model == "trans_unet":
        # reference main.py:93-101: patch_size is fixed at 4 on the command line; the 8-level default of
        # --channel-mults leaves a 1 x 1 bottleneck and no patches -- pass e.g. --channel-mults 1,2,2,4,4
        model = pai.TransUnetGAN(in_channels=1, out_channels=1, patch_size=4, channel_mults=channel_mults,
                                 dropout=hparams.dropout, loss_type=hparams.loss_type)
    elif hparams.model == "palette":
        raise NotImplementedError(
            "model 'palette' (the diffusion model) is outside the U-Net / Pix2Pix hot path this build covers "
            "(SURVEY.md section 8)")
    else:
        raise ValueError(f"Incorrect model name ({hparams.model})")

    rank, local, world = pdist.init_from_env()
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)

    if hparams.synthetic:
        # one data set (one seed) for the whole job; the module shards it across the ranks
        data_module = SyntheticDataModule(n_train=hparams.synthetic, n_val=max(hparams.batch_size, 8),
                                          batch_size=hparams.batch_size, size=hparams.image_size,
                                          seed=1234, world=world, rank=rank)
    else:
        # every rank reads its own shard of the lists (DistributedSampler semantics, as Lightning-DDP gives the
        # reference): an epoch is one pass over the data, not `world` passes
        data_module = ImageDataModule(hparams.data, hparams.val_data, batch_size=hparams.batch_size,
                                      normalize=True, num_workers=hparams.num_workers, world=world, rank=rank)

    checkpoint_callback = ModelCheckpoint(save_top_k=1, monitor="val_ssim", mode="max", filename="best",
                                          save_last=False)
    csv_logger = CSVLogger("logs", name=hparams.name)
    reducer = None
    if world > 1:
        model.to(device)
        pdist.broadcast_parameters(model)
        reducer = pdist.GradReducer()
    callbacks = [EMACallback(0.9999), checkpoint_callback] if hparams.ema else [checkpoint_callback]
    trainer = Trainer(max_epochs=hparams.epochs, max_steps=hparams.steps, log_every_n_steps=10,
                      check_val_every_n_epoch=hparams.val_epochs, logger=[csv_logger],
                      precision=hparams.precision, callbacks=callbacks, benchmark=True,
                      device=device, reducer=reducer)
    trainer.fit(model, data_module)


def build_parser():
    parser = ArgumentParser()
    parser.add_argument("name")
    parser.add_argument("-d", "--data", type=pathlib.Path,
                        help="YAML file containing filenames of images that make up the training data.")
    parser.add_argument("-vd", "--val-data", type=pathlib.Path,
                        help="YAML file containing filenames of images that make up the validation data.")
    parser.add_argument("-e", "--epochs", default=200, type=int)
    parser.add_argument("-s", "--steps", default=-1, type=int)
    parser.add_argument("--batch-size", default=8, type=int)
    parser.add_argument("--val-epochs", default=10, help="Validation run every n epochs.", type=int)
    parser.add_argument("--precision", default="32", help="Floating-point precision")
    parser.add_argument("--ema", default=False, action=argparse.BooleanOptionalAction,
                        help="Whether to use EMA weight updating.")
    parser.add_argument("--channel-mults", default="1,2,4,8,8,8,8,8",
                        help="Defines the U-net architecture's depth and width. Should be comma-separated "
                             "powers of 2.")
    parser.add_argument("--attention-res", default="8,4,2",
                        help="At what downsample multiples attention should be used, if the model supports it.")
    parser.add_argument("--dropout", default=0.0, type=float)
    parser.add_argument("--loss-type", default="gan", choices=["gan", "ssim", "psnr", "ssim+psnr", "mse"])
    parser.add_argument("--schedule-type", default="linear", choices=["linear", "cosine"])
    parser.add_argument("--learn-variance", default=False, action=argparse.BooleanOptionalAction)
    parser.add_argument("-m", "--model", default="pix2pix",
                        choices=["pix2pix", "attention_unet", "res18_unet", "res50_unet", "resv2_unet",
                                 "resnext_unet", "trans_unet", "palette"])
    # build-only additions
    parser.add_argument("--synthetic", default=0, type=int, help="train on N synthetic pairs")
    parser.add_argument("--image-size", default=256, type=int)
    parser.add_argument("--num-workers", default=0, type=int)
    return parser


if __name__ == "__main__":
    main(build_parser().parse_args())

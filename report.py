"""Evaluation CLI -- same flags and output tree as the reference's report.py:236-270
(reports/<name>/{depth_ssim.csv, outputs/, ssim_images/, stats.txt, ssim_per_image.csv,
psnr_per_image.csv, mse_per_image.csv}); the eval-mode generator forward and the SSIM / PSNR / MSE
arithmetic run on the MI355X kernels.

Reference defect handled here: report.py:152 counts FLOPs with a 3-channel input;
this counts the loaded model's own convolutions (1 "FLOP" per MAC, fvcore's convention).
"""
import os
import pathlib
from argparse import ArgumentParser

import numpy as np
import torch

import pai_bootstrap

pai = pai_bootstrap.load()
from thesis_pai_reconstruction_amd import functional as PF  # noqa: E402
from thesis_pai_reconstruction_amd.dataset import ImageDataModule, SyntheticDataModule  # noqa: E402
from thesis_pai_reconstruction_amd.models.utils import get_parameter_count, to_int  # noqa: E402


def write_png(img_u8: torch.Tensor, filename: str):
    """torchvision.io.write_png replacement: [C x H x W] uint8."""
    from PIL import Image
    a = img_u8.cpu().numpy()
    Image.fromarray(a[0] if a.shape[0] == 1 else np.transpose(a, (1, 2, 0))).save(filename)


def depth_ssim(preds: torch.Tensor, targets: torch.Tensor, num_depths: int = 16) -> torch.Tensor:
    """Mean / std of the per-image SSIM over `num_depths` horizontal strips (reference report.py:188-217)."""
    out = []
    for xp, xt in zip(preds.chunk(num_depths, dim=2), targets.chunk(num_depths, dim=2)):
        s = PF.ssim_per_image(xp.contiguous(), xt.contiguous())
        out.append((s.mean(), s.std()))
    return torch.tensor(out)


def output_hot_image(img: torch.Tensor, filename: str):
    """afmhot colormap PNG (reference report.py:220-233)."""
    from matplotlib import colormaps
    rgb = colormaps["afmhot"](img.cpu().numpy())[0, :, :, :3]
    write_png(to_int(torch.tensor(rgb, dtype=torch.float32).permute(2, 0, 1)), filename)


def count_macs(model) -> int:
    unet = getattr(model, "unet", None)
    if unet is not None and hasattr(unet, "vit_bottleneck"):      # TransUNet: convs at their output resolution + ViT
        def conv_macs(c, sp_out):
            return sp_out * sp_out * c.weight.shape[0] * c.weight.shape[1] * c.weight.shape[2] * c.weight.shape[3]
        size = unet.image_size
        total, sp = conv_macs(unet.in_conv, size), size
        for enc in unet.encoders:
            d = enc.decode
            total += conv_macs(d[0], sp) + conv_macs(d[3], sp // 2) + conv_macs(d[6], sp // 2) + conv_macs(enc.skip[0], sp // 2)
            sp //= 2
        vit = unet.vit_bottleneck
        P, D = vit.num_patches, vit.patch_dim
        per_token = D * D + sum(l.self_attn.in_proj_weight.numel() + l.self_attn.out_proj.weight.numel() +
                                l.linear1.weight.numel() + l.linear2.weight.numel() for l in vit.transformer.layers)
        total += P * per_token         # (attention over the batch axis adds 2 * N * D MACs per token, batch-dependent)
        for dec in unet.decoders:
            total += conv_macs(dec.decode[0], sp) + conv_macs(dec.decode[3], sp)
            sp *= 2
        return total + conv_macs(unet.out[0], sp)
    if unet is not None and hasattr(unet, "in_conv"):      # residual U-Net: every conv at the resolution it runs at
        def conv_macs(c, sp):
            return sp * sp * c.weight.shape[0] * c.weight.shape[1] * c.weight.shape[2] * c.weight.shape[3]
        size, total = 256, 0
        total += conv_macs(unet.in_conv, size)
        sp = size
        for enc in unet.encoders:
            total += sum(conv_macs(m, sp) for m in enc.encode[0].modules() if isinstance(m, torch.nn.Conv2d))
            sp //= 2
        for dec in unet.decoders:
            total += sum(conv_macs(m, sp) for m in dec.decode[0].modules() if isinstance(m, torch.nn.Conv2d))
            sp *= 2
        return total + conv_macs(unet.out[0], sp)
    if unet is None or not hasattr(unet, "engine"):
        return 0
    eng, size, total = unet.engine, 256, 0
    cin = eng.in_ch
    for i, c in enumerate(eng.enc_c):
        total += (size >> (i + 1)) ** 2 * 16 * cin * c
        cin = c
    for j, c in enumerate(eng.dec_c):
        hin = size >> (eng.L - j)
        k = eng.enc_c[-1] if j == 0 else eng.dec_c[j - 1] + eng.enc_c[eng.L - 1 - j]
        total += hin * hin * 16 * k * c
    for j, g in enumerate(getattr(eng, "gates", []), 1):      # attention gates: two C -> K pointwise convs, K -> 1
        sp = size >> (eng.L - j)
        total += sp * sp * (2 * g.C * g.K + g.K)
    return total


def main(hparams):
    dev = torch.device("cuda", 0)
    if hparams.model == "pix2pix":
        model = pai.Pix2Pix.load_from_checkpoint(hparams.checkpoint, map_location=dev)
        model.freeze()
    elif hparams.model == "attention_unet":
        model = pai.AttentionUnetGAN.load_from_checkpoint(hparams.checkpoint, map_location=dev)
        model.freeze()
    elif hparams.model in ("res18_unet", "res50_unet", "resv2_unet", "resnext_unet"):
        model = pai.ResUnetGAN.load_from_checkpoint(hparams.checkpoint, map_location=dev)
        model.freeze()
    elif hparams.model == "trans_unet":
        model = pai.TransUnetGAN.load_from_checkpoint(hparams.checkpoint, map_location=dev)
        model.freeze()
    elif hparams.model == "identity":
        def model(x):
            return x
    else:
        raise NotImplementedError(f"model {hparams.model!r} is not built on the HIP path yet")

    if hparams.data is None:
        data_module = SyntheticDataModule(n_val=16, batch_size=hparams.batch_size)
        data_module.setup("predict")
    else:
        data_module = ImageDataModule(hparams.data, batch_size=hparams.batch_size)
        data_module.setup("predict")
    dataloader = data_module.predict_dataloader()

    with torch.no_grad():
        preds = torch.cat([PF.denormalize(model(b[0].to(dev))) for b in dataloader], 0)
        targets = torch.cat([PF.denormalize(b[1].to(dev)) for b in dataloader], 0)

    ssims, ssim_images, psnrs, mses = [], [], [], []
    for pred, target in zip(preds.split(64), targets.split(64)):
        s, full = PF.ssim_per_image(pred, target, return_full_image=True)
        ssims.append(s)
        ssim_images.append(full)
        psnrs.append(torch.stack([PF.psnr(p[None], t[None]) for p, t in zip(pred, target)]))
        mses.append(torch.stack([PF.rmse(p[None], t[None]) ** 2 for p, t in zip(pred, target)]))
    ssims, ssim_images = torch.cat(ssims).cpu(), torch.cat(ssim_images).cpu()
    psnrs, mses = torch.cat(psnrs).cpu(), torch.cat(mses).cpu()

    ssim_over_depth = depth_ssim(preds, targets)
    report_dir = os.path.join("reports", hparams.name)
    os.makedirs(report_dir, exist_ok=True)
    with open(os.path.join(report_dir, "depth_ssim.csv"), "w") as f:
        f.write("depth,mean,std\n")
        for depth, (mean, std) in enumerate(ssim_over_depth, 1):
            f.write(f"{depth},{mean},{std}\n")
    outputs_dir = os.path.join(report_dir, "outputs")
    os.makedirs(outputs_dir, exist_ok=True)
    for index, pred in enumerate(preds.cpu()):
        output_hot_image(pred, os.path.join(outputs_dir, f"{str(index).zfill(5)}.png"))
    ssim_dir = os.path.join(report_dir, "ssim_images")
    os.makedirs(ssim_dir, exist_ok=True)
    for index, img in enumerate(ssim_images):
        write_png(to_int(img.clamp(0, 1)), os.path.join(ssim_dir, f"{str(index).zfill(5)}.png"))

    rmse_stat = PF.rmse(preds, targets)
    with open(os.path.join(report_dir, "stats.txt"), "w") as f:
        f.write(f"SSIM: {ssims.mean()}\n")
        f.write(f"PSNR: {psnrs.mean()}\n")
        f.write(f"RMSE: {float(rmse_stat)}\n")
        f.write(f"FLOPs: {count_macs(model) if isinstance(model, torch.nn.Module) else 0}\n")
        f.write(f"Parameter count: {get_parameter_count(model)}\n")
    for fname, header, vals in (("ssim_per_image.csv", "image,ssim", ssims),
                                ("psnr_per_image.csv", "image,psnr", psnrs),
                                ("mse_per_image.csv", "image,mse", mses)):
        with open(os.path.join(report_dir, fname), "w") as f:
            f.write(header + "\n")
            for index, v in enumerate(vals):
                f.write(f"{str(index).zfill(5)},{v}\n")


def build_parser():
    parser = ArgumentParser()
    parser.add_argument("name")
    parser.add_argument("-c", "--checkpoint", type=pathlib.Path, help="Path to checkpoint")
    parser.add_argument("-d", "--data", type=pathlib.Path, help="YAML file of all data points")
    parser.add_argument("-bs", "--batch-size", default=2, type=int)
    parser.add_argument("-m", "--model", default="pix2pix",
                        choices=["pix2pix", "attention_unet", "res18_unet", "res50_unet", "resv2_unet",
                                 "resnext_unet", "trans_unet", "palette", "identity"])
    return parser


if __name__ == "__main__":
    main(build_parser().parse_args())

"""MI355X-native Pix2Pix / U-Net training hot path (drop-in for the ``UnetWrapper`` /
``Discriminator`` plugin surface of cristianpjensen/thesis-pai-reconstruction).

Import name: ``thesis_pai_reconstruction_amd`` (via ``pai_bootstrap.load()``; the directory
name carries a hyphen).  All arithmetic lives in ``libpai_hip.so`` (csrc/, C ABI in
include/pai_hip.h); this package holds the host-side mirror of the reference interface.
"""
from . import lib  # noqa: F401
from .lib import PaiError  # noqa: F401
from .models.pix2pix import Pix2Pix, Unet, EncoderBlock, DecoderBlock  # noqa: F401
from .models.attention_unet import AttentionUnetGAN, AttentionUnet, AttentionBlock  # noqa: F401
from .models.res_unet import ResUnetGAN, ResUnet  # noqa: F401
from .models.trans_unet import TransUnetGAN, TransUnet  # noqa: F401
from .models.wrapper import UnetWrapper, Discriminator, DiscriminatorBlock  # noqa: F401
from .lightning import Trainer, CSVLogger, ModelCheckpoint, LightningModule  # noqa: F401

__all__ = ["Pix2Pix", "Unet", "AttentionUnetGAN", "AttentionUnet", "ResUnetGAN", "ResUnet", "TransUnetGAN", "TransUnet", "UnetWrapper", "Discriminator", "Trainer", "CSVLogger", "ModelCheckpoint",
           "PaiError", "lib"]

"""CPU oracle for the Pix2Pix / U-Net training hot path.

TEST INFRASTRUCTURE ONLY.  This package is a CPU restatement (PyTorch-CPU fp32
ops + explicit numpy/torch formulas for the third-party metric code) of the
algorithm that cristianpjensen/thesis-pai-reconstruction executes on its
``UnetWrapper.training_step`` path.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it, and only as the checker
or the timed CPU baseline -- never as a product code path.

Pinning status (see DESIGN.md "Oracle"):
  * model / loss / optimiser-toggle / BN double-update semantics are pinned
    against the REAL reference classes, imported in the build container under
    the shims in ``oracle/shims`` by ``oracle/gen_golden.py``; the outputs are
    committed as ``tests/golden/*.npz`` and ``tests/test_oracle_golden.py``
    replays them against this restatement.
  * SSIM is pinned against an independent implementation (scikit-image 0.18.3,
    ``oracle/gen_ssim_skimage.py``); torchmetrics 0.11.4 itself is not
    installed anywhere in this environment, so the un-cropped
    ``return_full_image`` border of the SSIM map is "parity unpinned".
"""

from .pix2pix_ref import (  # noqa: F401
    make_unet_state, make_disc_state, unet_forward, disc_forward,
    init_state_portable, dropout_rates,
)
from .attention_ref import make_attention_unet_state, attention_unet_forward, attention_block  # noqa: F401
from .res_unet_ref import make_res_unet_state, res_unet_forward  # noqa: F401
from .trans_unet_ref import make_trans_unet_state, trans_unet_forward, init_trans_state_portable  # noqa: F401
from .metrics_ref import denormalize, ssim, ssim_full, psnr, rmse, mse  # noqa: F401
from .step_ref import (  # noqa: F401
    AdamState, adam_step, gan_training_step, plain_training_step,
    generator_loss, discriminator_loss, validation_step,
)

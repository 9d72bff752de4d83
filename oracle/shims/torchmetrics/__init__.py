"""Stand-in for torchmetrics 0.11.4 (not installed).  The three functionals are
RESTATED from the published 0.11.4 algorithm in oracle/metrics_ref.py; SSIM is
independently pinned against scikit-image.  TEST INFRASTRUCTURE ONLY."""
from . import functional  # noqa: F401

"""Import-time stand-in for torchvision (not installed); only what
models/utils.py:3,11-12 touches.  TEST INFRASTRUCTURE ONLY."""
from . import transforms, io  # noqa: F401

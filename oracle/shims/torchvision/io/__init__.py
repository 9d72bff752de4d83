class ImageReadMode:
    GRAY = 1


def read_image(*a, **k):
    raise RuntimeError("shim")


def write_png(*a, **k):
    raise RuntimeError("shim")

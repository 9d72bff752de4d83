import torch
from oracle import metrics_ref as _m


def structural_similarity_index_measure(preds, target, data_range=None,
                                        return_full_image=False, reduction="elementwise_mean"):
    per_image, full = _m.ssim_full(preds, target, data_range=1.0 if data_range is None else data_range)
    val = per_image.mean() if reduction == "elementwise_mean" else per_image
    if return_full_image:
        return val, full
    return val


def peak_signal_noise_ratio(preds, target, data_range=None):
    return _m.psnr(preds, target, data_range=1.0 if data_range is None else data_range)


def mean_squared_error(preds, target, squared=True):
    v = _m.mse(preds, target)
    return v if squared else torch.sqrt(v)

"""Image-error figures of the parity bar: per-pixel L2 normalised by the mean reference luminance,
and the reference tooling's relMSE."""
import numpy as np


def l2_over_luminance(img, ref, lum=None):
    """sqrt(mean((img - ref)^2)) / mean reference luminance (the throughput triple of `ref` when it has 27 planes)."""
    img = np.asarray(img, np.float64)
    ref = np.asarray(ref, np.float64)
    if lum is None:
        lum = ref[..., 0:3].mean()
    return float(np.sqrt(((img - ref) ** 2).mean()) / max(lum, 1e-300))


def rel_mse(img, ref):
    """ERelMSE of scripts/rgbe/sources/imageerrors.h:81-87,117-121,135-144 (metricPix + errorNorm):
    per pixel diff = mean_c(img_c - ref_c); refGray = mean_c(ref_c); error = mean_pixels(diff^2 / (refGray^2 + 0.001)).
    img, ref: (..., 3) RGB images."""
    img = np.asarray(img, np.float64).reshape(-1, 3)
    ref = np.asarray(ref, np.float64).reshape(-1, 3)
    if img.shape[0] == 0:
        return 0.0
    diff = (img - ref).sum(1) / 3.0
    gray = ref.sum(1) / 3.0
    return float((diff * diff / (gray * gray + 0.001)).mean())

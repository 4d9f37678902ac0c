"""Host-side image normalisation (reference: empanada_napari/utils.py:153-201).

``Preprocessor`` keeps the reference's contract: integer-typed 2-D numpy image
in, ``{'image': FloatTensor (1,H,W)}`` out, float input rejected with the same
exception.  The GPU engine can also take the raw integer tile and fuse the
same arithmetic into its stem kernel (``normalize_params`` gives it the two
constants), which is the path ``bench.py`` times.
"""
import numpy as np


def normalize_params(mean, std, max_pixel_value):
    """-> (sub, mul) fp32 such that out = (img - sub) * mul  (utils.py:153-165)."""
    m = np.float32(np.array(mean, dtype=np.float32) * max_pixel_value)
    s = np.float32(np.array(std, dtype=np.float32) * max_pixel_value)
    return m, np.reciprocal(s, dtype=np.float32)


def normalize(img, mean, std, max_pixel_value=255.0):
    sub, mul = normalize_params(mean, std, max_pixel_value)
    out = img.astype(np.float32)
    out -= sub
    out *= mul
    return out


class Preprocessor:
    def __init__(self, mean=None, std=None):
        self.mean = mean
        self.std = std

    def __call__(self, image=None):
        import torch
        assert image is not None
        if np.issubdtype(image.dtype, np.floating):
            raise Exception('Input image cannot be float type!')
        max_value = np.iinfo(image.dtype).max
        image = normalize(image, self.mean, self.std, max_pixel_value=max_value)
        return {'image': torch.from_numpy(image[None])}


def resize_by_factor(image, scale_factor=1):
    """data/utils/transforms.py:9-21: cv2.resize(image, (ceil(w/s), ceil(h/s))) with cv2's default
    INTER_LINEAR (half-pixel centres, edge replicate, no anti-aliasing) == bilinear, align_corners=False.
    Pinned by the reference's three float cases (tests/test_transforms.py:6-28, restated in
    tests/test_host_misc.py).  Integer images: cv2 (opencv-python >= 4.11, absent from this image) works in fixed
    point.  Its published arithmetic (imgproc/resize.cpp) for the cases this path allows -- power-of-two factors,
    volume_dataset.py:26-27 -- on sides divisible by the factor: factor 2 is re-routed to the fast area path,
    dst = (a + b + c + d + 2) >> 2 over the 2x2 block; factors 4, 8 sample at 4k + 1.5 (weights exactly 1/2, 11-bit
    coefficients 1024 + 1024), and the 8-bit vertical pass ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2 reduces
    to the same (a + b + c + d + 2) >> 2 over the four centre pixels.  floor(mean + 0.5) below is that value, so uint8
    images with sides divisible by the factor agree by arithmetic; odd sides (fractional weights rounded to 11 bits)
    and uint16 at factors > 2 (cv2 rounds a float result half to even) remain PARITY UNPINNED."""
    if scale_factor == 1:
        return image
    import math
    import torch
    import torch.nn.functional as F
    h, w = image.shape
    dh, dw = math.ceil(h / scale_factor), math.ceil(w / scale_factor)
    x = torch.from_numpy(np.ascontiguousarray(image).astype(np.float64))[None, None]
    y = F.interpolate(x, size=(dh, dw), mode='bilinear', align_corners=False)[0, 0].numpy()
    if np.issubdtype(image.dtype, np.integer):
        info = np.iinfo(image.dtype)
        y = np.clip(np.floor(y + 0.5), info.min, info.max)
    return y.astype(image.dtype)

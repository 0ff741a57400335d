"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the image transform chain of the finetune drivers.

Reference: moco_pretraining/moco/aihc_utils/image_transform.py:50-84 (`get_transform_type`): Resize((S, S)) ->
RandomHorizontalFlip -> RandomRotation(rotate) -> RandomCrop((crop, crop)) | CenterCrop -> ToTensor -> Normalize(mean, std), applied
to `Image.fromarray(cv2.imread(path))` (moco/loader.py:121-125: BGR channel order kept as is).  The arithmetic lives in torchvision
(absent here, unpinned) on top of Pillow (installed): torchvision's PIL backend forwards to
  Resize            -> Image.resize((w, h), BILINEAR)              (Pillow's antialiased separable resample, 8-bit fixed point)
  hflip             -> Image.transpose(FLIP_LEFT_RIGHT)
  rotate            -> Image.rotate(angle, NEAREST, expand=False, center=None, fillcolor=0)
  crop              -> Image.crop((j, i, j + w, i + h))
  ToTensor          -> uint8 HWC -> float32 CHW / 255
  Normalize         -> (x - mean) / std per channel
The Pillow algorithms (libImaging Resample.c / Geometry.c, Image.rotate) are restated here in numpy with their integer arithmetic;
tests/test_oracle_golden.py pins every function against the installed Pillow on seeded images, bit for bit.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bilinear(x):
    x = abs(x)
    return 1.0 - x if x < 1.0 else 0.0


def resample_coeffs(in_size, out_size):
    """Pillow precompute_coeffs + normalize_coeffs_8bpc for the BILINEAR filter over the whole axis.
    Returns (ksize, bounds int32 [out][2] = (xmin, count), kk int32 [out][ksize])."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_bilinear((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = sum(w)                                                        # left-to-right double sum, as the C loop
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def _clip8(v):
    return np.clip(v >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resize_bilinear_u8(img, out_h, out_w):
    """Image.resize((out_w, out_h), BILINEAR) of a uint8 HWC image: horizontal pass to uint8, then vertical pass."""
    in_h, in_w, _ = img.shape
    src = img.astype(np.int64)
    if out_w != in_w:
        _, bx, kx = resample_coeffs(in_w, out_w)
        tmp = np.empty((in_h, out_w, img.shape[2]), dtype=np.uint8)
        for xx in range(out_w):
            x0, n = bx[xx]
            acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(src[:, x0:x0 + n, :], kx[xx, :n].astype(np.int64), axes=([1], [0]))
            tmp[:, xx, :] = _clip8(acc)
        src = tmp.astype(np.int64)
    if out_h != in_h:
        _, by, ky = resample_coeffs(in_h, out_h)
        out = np.empty((out_h, src.shape[1], img.shape[2]), dtype=np.uint8)
        for yy in range(out_h):
            y0, n = by[yy]
            acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(src[y0:y0 + n, :, :], ky[yy, :n].astype(np.int64), axes=([0], [0]))
            out[yy] = _clip8(acc)
        return out
    return src.astype(np.uint8)


def hflip(img):
    return img[:, ::-1, :].copy()


def rotate_matrix(angle, w, h):
    """The 6 affine coefficients Image.rotate hands to transform(AFFINE) (output -> input pixel map), or None for its fast paths."""
    angle = angle % 360.0
    if angle == 0:
        return None
    if angle == 180 or (angle in (90, 270) and w == h):
        return ("transpose", angle)
    cx, cy = w / 2, h / 2
    a = -math.radians(angle)
    m = [round(math.cos(a), 15), round(math.sin(a), 15), 0.0, round(-math.sin(a), 15), round(math.cos(a), 15), 0.0]
    m[2] = m[0] * (-cx) + m[1] * (-cy) + m[2] + cx
    m[5] = m[3] * (-cx) + m[4] * (-cy) + m[5] + cy
    return m


def affine_fixed_params(m):
    """libImaging affine_fixed: 16.16 fixed-point increments and the start values of the pixel-centre map."""
    fix = lambda v: int(math.floor(v * 65536.0 + 0.5))
    a0, a1, a3, a4 = fix(m[0]), fix(m[1]), fix(m[3]), fix(m[4])
    a2 = fix(m[2] + m[0] * 0.5 + m[1] * 0.5)
    a5 = fix(m[5] + m[3] * 0.5 + m[4] * 0.5)
    return a0, a1, a2, a3, a4, a5


def rotate_nearest(img, angle, fill=0):
    """Image.rotate(angle, NEAREST, expand=False, fillcolor=fill) of a uint8 HWC image."""
    h, w, _ = img.shape
    m = rotate_matrix(angle, w, h)
    if m is None:
        return img.copy()
    if isinstance(m, tuple):
        k = {90: 1, 180: 2, 270: 3}[int(m[1])]
        return np.rot90(img, k).copy()                                     # ROTATE_90 is counter-clockwise, like np.rot90
    a0, a1, a2, a3, a4, a5 = affine_fixed_params(m)
    ys, xs = np.mgrid[0:h, 0:w].astype(np.int64)
    xin = (a2 + a1 * ys + a0 * xs) >> 16
    yin = (a5 + a4 * ys + a3 * xs) >> 16
    ok = (xin >= 0) & (xin < w) & (yin >= 0) & (yin < h)
    out = np.full_like(img, fill)
    out[ok] = img[yin[ok], xin[ok]]
    return out


def crop(img, i, j, th, tw):
    return img[i:i + th, j:j + tw, :].copy()


def center_crop_offsets(h, w, th, tw):
    """torchvision CenterCrop: int(round((h - th) / 2.0)), int(round((w - tw) / 2.0))  (Python round: half to even)."""
    return int(round((h - th) / 2.0)), int(round((w - tw) / 2.0))


def to_tensor_normalize(img, mean, std):
    x = img.astype(np.float32).transpose(2, 0, 1) / np.float32(255.0)
    mean = np.asarray(mean, dtype=np.float32).reshape(-1, 1, 1)
    std = np.asarray(std, dtype=np.float32).reshape(-1, 1, 1)
    return (x - mean) / std


def transform(img, size, flip, angle, crop_ij, crop_size, mean, std):
    """The whole chain of image_transform.py:50-84 for ONE decoded uint8 HWC image with the random draws given."""
    x = resize_bilinear_u8(img, size, size)
    if flip:
        x = hflip(x)
    x = rotate_nearest(x, angle)
    if crop_size:
        x = crop(x, crop_ij[0], crop_ij[1], crop_size, crop_size)
    return to_tensor_normalize(x, mean, std)


def transform_mocov3(img, box, size, flip, angle, mean, std):
    """Training chain of image_transform.py:86-95 (`get_transform_type_mocov3`): RandomResizedCrop = crop the drawn box (i, j, h, w),
    resize it to (size, size) (torchvision resized_crop = crop then resize), flip, rotate; then ToTensor + Normalize."""
    i, j, h, w = box
    x = resize_bilinear_u8(crop(img, i, j, h, w), size, size)
    if flip:
        x = hflip(x)
    x = rotate_nearest(x, angle)
    return to_tensor_normalize(x, mean, std)

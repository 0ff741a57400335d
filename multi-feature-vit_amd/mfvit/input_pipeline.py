"""GPU-side input pipeline: the finetune transform chain of aihc_utils/image_transform.py:50-84 as one HIP kernel.

    tf = GpuTransform(img_type="CheXpert-v1.0-small", img_size=256, crop=224, rotate=10, training=True)
    batch = tf(list_of_uint8_HWC_arrays)            # float32 [B, 3, 224, 224] on the GPU, what the DataLoader used to deliver

The DataLoader workers then only decode (cv2.imread, moco/loader.py:121) and hand over uint8 HWC arrays of any size; Resize((S,S))
-> RandomHorizontalFlip -> RandomRotation(rotate) -> RandomCrop((crop,crop)) | CenterCrop -> ToTensor -> Normalize run fused on
the device, bit-exact against Pillow's integer arithmetic (the backend torchvision's PIL transforms call).  This module is the
host half: the per-axis fixed-point coefficient tables and the 16.16 affine terms, computed in double exactly as Pillow does.
"""
import ctypes
import math

import numpy as np
import torch

from . import _lib
from ._lib import check, lib, ptr, stream

# per-dataset statistics, image_transform.py:4-19
NORMALIZE = {
    "CheXpert-v1.0-small": ([.5020, .5020, .5020], [float(np.round(np.sqrt(.085585), 4))] * 3),
    "CheXpert_Enh": ([.6086, .5204, .3384], [.134909, .088268, .035044]),
    "data": ([0.5045, 0.5045, 0.5045], [0.2462, 0.2462, 0.2462]),
    "Train_Mix": ([0.2243, 0.5507, 0.6865], [0.1026, 0.2995, 0.3300]),
}
_PREC = 32 - 8 - 2
_AXIS_CACHE = {}


def axis_table(in_size, out_size):
    """int32 [out_size][2 + ksize]: first source index, tap count, taps - Pillow's precompute_coeffs + normalize_coeffs_8bpc for
    the triangle (BILINEAR) filter stretched by max(scale, 1) (antialiasing on downscale)."""
    key = (in_size, out_size)
    if key in _AXIS_CACHE:
        return _AXIS_CACHE[key]
    scale = float(in_size) / out_size
    fscale = scale if scale > 1.0 else 1.0
    support = fscale
    ksize = int(math.ceil(support)) * 2 + 1
    tab = np.zeros((out_size, 2 + ksize), dtype=np.int32)
    inv = 1.0 / fscale
    for o in range(out_size):
        center = (o + 0.5) * scale
        lo = max(int(center - support + 0.5), 0)
        hi = min(int(center + support + 0.5), in_size)
        n = hi - lo
        taps, total = [], 0.0
        for t in range(n):
            u = abs((t + lo - center + 0.5) * inv)
            wgt = 1.0 - u if u < 1.0 else 0.0
            taps.append(wgt)
            total += wgt
        tab[o, 0], tab[o, 1] = lo, n
        for t in range(n):
            v = taps[t] / total if total != 0.0 else taps[t]
            tab[o, 2 + t] = int(v * (1 << _PREC) - 0.5) if v < 0 else int(v * (1 << _PREC) + 0.5)
    _AXIS_CACHE[key] = (ksize, tab)
    return ksize, tab


def rotation_terms(angle, size):
    """(mode, a0..a5): Image.rotate(angle, NEAREST, expand=False) on a size x size image.  mode 0 none, 1 affine (16.16 fixed-point
    terms of libImaging's affine_fixed), 2/3/4 the transpose fast paths for 90/180/270 degrees."""
    angle = angle % 360.0
    if angle == 0:
        return 0, (0,) * 6
    if angle in (90, 180, 270):
        return {90: 2, 180: 3, 270: 4}[int(angle)], (0,) * 6
    c = size / 2
    a = -math.radians(angle)
    m = [round(math.cos(a), 15), round(math.sin(a), 15), 0.0, round(-math.sin(a), 15), round(math.cos(a), 15), 0.0]
    m[2] = m[0] * (-c) + m[1] * (-c) + m[2] + c
    m[5] = m[3] * (-c) + m[4] * (-c) + m[5] + c
    fix = lambda v: int(math.floor(v * 65536.0 + 0.5))
    return 1, (fix(m[0]), fix(m[1]), fix(m[2] + m[0] * 0.5 + m[1] * 0.5), fix(m[3]), fix(m[4]), fix(m[5] + m[3] * 0.5 + m[4] * 0.5))


class GpuTransform:
    """Mirror of `get_transform_type(args, training, img_type)` (image_transform.py:50-84) with args.maintain_ratio False:
    args.img_size -> img_size, args.crop -> crop (0 = no crop), args.rotate -> rotate (degrees; RandomRotation draws from
    [-rotate, rotate])."""

    def __init__(self, img_type="CheXpert-v1.0-small", img_size=256, crop=224, rotate=10, training=True, device="cuda:0",
                 mocov3=False, crop_min=0.08):
        """mocov3=True mirrors `get_transform_type_mocov3` (image_transform.py:86-124, MoCo pretraining): training =
        RandomResizedCrop(img_size, scale=(crop_min, 1)) -> flip -> rotation (no further crop); evaluation = Resize((256, 256)) ->
        CenterCrop(crop)."""
        self.mocov3, self.crop_min = bool(mocov3), float(crop_min)
        if mocov3:
            if training:
                crop = 0
            else:
                img_size = 256
        if img_type not in NORMALIZE:
            raise _lib.MfvitError(f"unknown img_type {img_type!r} (image_transform.py:72-81 knows {sorted(NORMALIZE)})")
        self.mean, self.std = NORMALIZE[img_type]
        self.size, self.crop, self.rotate, self.training = int(img_size), int(crop) if crop else int(img_size), float(rotate), training
        if self.crop > self.size:
            raise _lib.MfvitError("crop larger than the resized image")
        self.device = torch.device(device)

    @staticmethod
    def resized_crop_box(height, width, scale, generator=None, ratio=(3.0 / 4.0, 4.0 / 3.0)):
        """torchvision RandomResizedCrop.get_params: (i, j, h, w) of the source window (10 tries, then the central fallback)."""
        area = height * width
        log_ratio = torch.log(torch.tensor(ratio))
        for _ in range(10):
            target_area = area * torch.empty(1).uniform_(scale[0], scale[1], generator=generator).item()
            aspect = torch.exp(torch.empty(1).uniform_(float(log_ratio[0]), float(log_ratio[1]), generator=generator)).item()
            w = int(round(math.sqrt(target_area * aspect)))
            h = int(round(math.sqrt(target_area / aspect)))
            if 0 < w <= width and 0 < h <= height:
                i = int(torch.randint(0, height - h + 1, (1,), generator=generator))
                j = int(torch.randint(0, width - w + 1, (1,), generator=generator))
                return i, j, h, w
        in_ratio = float(width) / float(height)
        if in_ratio < min(ratio):
            w = width
            h = int(round(w / min(ratio)))
        elif in_ratio > max(ratio):
            h = height
            w = int(round(h * max(ratio)))
        else:
            w, h = width, height
        return (height - h) // 2, (width - w) // 2, h, w

    def sample_params(self, n, generator=None, sizes=None):
        """The random draws of one batch, in torchvision's order per image: [mocov3: the RandomResizedCrop box, needs `sizes` =
        [(h, w)] of the images], flip (torch.rand(1) < 0.5), angle (uniform in [-rotate, rotate]), crop offsets (randint);
        evaluation: no flip, no rotation, CenterCrop offsets.  Tuples (flip, angle, crop_i, crop_j[, box])."""
        S, C = self.size, self.crop
        out = []
        for s in range(n):
            if self.training and self.mocov3:
                box = self.resized_crop_box(sizes[s][0], sizes[s][1], (self.crop_min, 1.0), generator)
                flip = bool(torch.rand(1, generator=generator) < 0.5)
                angle = float(torch.empty(1).uniform_(-self.rotate, self.rotate, generator=generator))
                out.append((flip, angle, 0, 0, box))
                continue
            if self.training:
                flip = bool(torch.rand(1, generator=generator) < 0.5)
                angle = float(torch.empty(1).uniform_(-self.rotate, self.rotate, generator=generator))
                i = int(torch.randint(0, S - C + 1, (1,), generator=generator))
                j = int(torch.randint(0, S - C + 1, (1,), generator=generator))
            else:
                flip, angle = False, 0.0
                i = j = int(round((S - C) / 2.0))
            out.append((flip, angle, i, j))
        return out

    def __call__(self, images, params=None, generator=None):
        """images: list of uint8 HWC (3-channel) numpy arrays / CPU tensors of any size.  Returns float32 [n, 3, crop, crop] on
        the device.  params: list of (flip, angle, crop_i, crop_j) per image (default: sample_params)."""
        if not torch.cuda.is_available():
            raise _lib.MfvitError("GpuTransform needs the GPU (no CPU fallback)")
        n = len(images)
        S, C = self.size, self.crop
        arrs = []
        for im in images:
            a = im.numpy() if isinstance(im, torch.Tensor) else np.asarray(im)
            if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
                raise _lib.MfvitError("images must be uint8 HWC with 3 channels (Image.fromarray(cv2.imread(..)), loader.py:121-125)")
            arrs.append(np.ascontiguousarray(a))
        if params is None:
            params = self.sample_params(n, generator, [a.shape[:2] for a in arrs])
        desc = np.zeros((n, 20), dtype=np.int64)
        tabs, tab_off, tab_pos, off = [], {}, 0, 0
        for s, (a, prm) in enumerate(zip(arrs, params)):
            flip, angle, ci, cj = prm[:4]
            H, W = a.shape[:2]
            bi, bj, h, w = prm[4] if len(prm) > 4 else (0, 0, H, W)      # source window (RandomResizedCrop box) or the whole image
            if not (0 <= bi and 0 <= bj and h > 0 and w > 0 and bi + h <= H and bj + w <= W):
                raise _lib.MfvitError("source window outside the image")
            for axis, size in ((0, w), (1, h)):
                if size not in tab_off:
                    ks, t = axis_table(size, S)
                    tab_off[size] = (tab_pos, ks)
                    tabs.append(t.reshape(-1))
                    tab_pos += t.size
            mode, terms = rotation_terms(angle, S)
            if not (0 <= ci <= S - C and 0 <= cj <= S - C):
                raise _lib.MfvitError("crop offset out of range")
            desc[s] = [off + (bi * W + bj) * 3, h, w, tab_off[w][0], tab_off[h][0], tab_off[w][1], tab_off[h][1], int(flip), mode, *terms,
                       (ci << 32) | cj, W * 3, 0, 0, 0]
            off += a.size
        src = torch.from_numpy(np.concatenate([a.reshape(-1) for a in arrs])).to(self.device, non_blocking=True)
        dsc = torch.from_numpy(desc).to(self.device, non_blocking=True)
        tab = torch.from_numpy(np.concatenate(tabs)).to(self.device, non_blocking=True)
        out = torch.empty(n, 3, C, C, device=self.device, dtype=torch.float32)
        mean = (ctypes.c_float * 3)(*self.mean)
        std = (ctypes.c_float * 3)(*self.std)
        check(lib().mfvit_input_transform(ptr(src), ptr(dsc), ptr(tab), n, S, C, ctypes.cast(mean, ctypes.c_void_p),
                                          ctypes.cast(std, ctypes.c_void_p), ptr(out), stream()), "mfvit_input_transform")
        return out

"""load_and_preprocess_images on the GPU -- reference iggt/utils/load_fn.py:12-128 (caller: demo.py:182).

Same signature, modes ("crop" / "pad" / "resize"), target sizes (518, multiples of 14), white padding and output
([N, 3, H, W] fp32 in [0, 1]) as the reference; the result is bit-identical to the reference's
PIL.Image.resize(BICUBIC) + torchvision ToTensor path (tests/test_preprocess_gpu.py), but only the JPEG/PNG decode (PIL)
runs on the host: the resize is Pillow's integer resampler re-implemented as two HIP kernels
(csrc/preprocess.hip), ToTensor + centre crop + padding one more, and the images land directly in the [N, 3, H, W] device
tensor the forward takes.  The coefficient tables are built here exactly as Pillow's precompute_coeffs() /
normalize_coeffs_8bpc() do (double precision, same operation order) and cached per (in size, out size)."""
import math
from functools import lru_cache

import torch
from PIL import Image

from .. import _C

_DEFAULT = 518
_PRECISION_BITS = 32 - 8 - 2


def _bicubic(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


@lru_cache(maxsize=64)
def _coeffs(in_size: int, out_size: int):
    """(bounds int32 [out,2], kk int32 [out,ksize]) of Pillow's 8-bit bicubic resampler for the full-image box."""
    scale = in_size / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = torch.zeros(out_size, 2, dtype=torch.int32)
    kk = torch.zeros(out_size, ksize, dtype=torch.int32)
    ss = 1.0 / filterscale
    one = float(1 << _PRECISION_BITS)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for w in k:
            ww += w
        if ww != 0.0:
            k = [w / ww for w in k]
        bounds[xx, 0], bounds[xx, 1] = xmin, xmax
        for x, w in enumerate(k):
            kk[xx, x] = int(-0.5 + w * one) if w < 0 else int(0.5 + w * one)
    return bounds, kk


_DEV_TABLES = {}


def _tables(in_size, out_size, device):
    key = (in_size, out_size, str(device))
    if key not in _DEV_TABLES:
        b, k = _coeffs(in_size, out_size)
        _DEV_TABLES[key] = (b.to(device), k.to(device))
    return _DEV_TABLES[key]


def _open_rgb(path):
    img = Image.open(path)
    if img.mode == "RGBA":   # blend onto white (load_fn.py:62-64)
        background = Image.new("RGBA", img.size, (255, 255, 255, 255))
        img = Image.alpha_composite(background, img)
    return img.convert("RGB")


def _resized_u8(img: Image.Image, new_w: int, new_h: int, device):
    """uint8 [new_h, new_w, 3] on `device`, identical to img.resize((new_w, new_h), BICUBIC)."""
    w, h = img.size
    src = torch.frombuffer(bytearray(img.tobytes()), dtype=torch.uint8).view(h, w, 3).to(device, non_blocking=True)
    if (w, h) == (new_w, new_h):
        return src                                   # Pillow returns a copy without resampling
    if w == new_w or h == new_h:
        # Pillow skips the pass whose size does not change; the tables of an identity pass reproduce the input exactly
        pass
    hb, hk = _tables(w, new_w, device)
    vb, vk = _tables(h, new_h, device)
    return _C.resize_bicubic_u8(src, hb, hk, vb, vk, new_h, new_w)


def load_and_preprocess_images(image_path_list, mode="crop", resize_target_size=None, device="cuda"):
    """List of image paths -> [N, 3, H, W] fp32 tensor in [0, 1] on `device` (a ROCm device)."""
    if not image_path_list:
        raise ValueError("At least 1 image is required")
    if mode not in ["crop", "pad", "resize"]:
        raise ValueError("Mode must be either 'crop', 'pad', or 'resize'")
    if mode == "resize":
        if resize_target_size is None:
            raise ValueError("resize_target_size must be provided as a (width, height) tuple when mode is 'resize'")
        if not (isinstance(resize_target_size, (tuple, list)) and len(resize_target_size) == 2):
            raise ValueError("resize_target_size must be a tuple or list of two integers: (width, height)")
    device = torch.device(device)
    if device.type != "cuda":
        raise _C.HipExtensionError("load_and_preprocess_images resizes on the GPU: pass a ROCm device (no CPU fallback)")
    _C.load()

    frames = []   # (uint8 HWC tensor, crop_y, h, w, H_final, W_final, pad_y, pad_x)
    for path in image_path_list:
        img = _open_rgb(path)
        width, height = img.size
        if mode == "pad":
            if width >= height:
                new_w = _DEFAULT
                new_h = round(height * (new_w / width) / 14) * 14
            else:
                new_h = _DEFAULT
                new_w = round(width * (new_h / height) / 14) * 14
        elif mode == "resize":
            new_w, new_h = resize_target_size
        else:
            new_w = _DEFAULT
            new_h = round(height * (new_w / width) / 14) * 14
        u8 = _resized_u8(img, int(new_w), int(new_h), device)
        crop_y, h, w = 0, int(new_h), int(new_w)
        Hf, Wf, py, px = h, w, 0, 0
        if mode == "crop" and new_h > _DEFAULT:
            crop_y, h, Hf = (new_h - _DEFAULT) // 2, _DEFAULT, _DEFAULT
        elif mode == "pad":
            Hf, Wf = max(h, _DEFAULT), max(w, _DEFAULT)
            py, px = (Hf - h) // 2, (Wf - w) // 2
        frames.append([u8, crop_y, h, w, Hf, Wf, py, px])

    shapes = {(f[4], f[5]) for f in frames}
    Hmax, Wmax = max(s[0] for s in shapes), max(s[1] for s in shapes)
    if len(shapes) > 1:
        print(f"Warning: Found images with different shapes after processing: {shapes}")
    out = torch.empty(len(frames), 3, Hmax, Wmax, dtype=torch.float32, device=device)
    for i, (u8, crop_y, h, w, Hf, Wf, py, px) in enumerate(frames):
        # second-level padding to the common size (load_fn.py:104-121) composes with the first: both are centred white pads
        py2, px2 = (Hmax - Hf) // 2, (Wmax - Wf) // 2
        _C.u8hwc_to_f32chw(u8, out[i], crop_y, 0, py + py2, px + px2, h, w, 1.0)
    return out

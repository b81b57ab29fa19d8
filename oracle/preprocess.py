"""CPU restatement of the reference's input transform (TEST INFRASTRUCTURE, see oracle/__init__.py):

    CenterCrop((178,178)) -> Resize((128,128)) -> RandomHorizontalFlip -> ToTensor -> MinMax(True)
    (05-train notebook cell 9; MinMax / min_max: pyfiles/util.py:108-155; FaceDataset.__getitem__: pyfiles/dataset.py:127-140)

torchvision is a third-party dependency that is absent from this image; its PIL code path is three Pillow calls
(``Image.crop``, ``Image.resize(size, BILINEAR)``, ``Image.transpose(FLIP_LEFT_RIGHT)``), which ``transform_pil`` makes
directly -- Pillow IS installed, so this oracle is pinned by running the real library.  ``resize_restated`` repeats Pillow's
two-pass fixed-point resample (libImaging/Resample.c) in numpy from the same tables the HIP path uploads; tests hold it
bit-exact against Pillow, which pins the tables themselves."""
import numpy as np


def center_crop_box(h, w, ch, cw):
    return int(round((h - ch) / 2.0)), int(round((w - cw) / 2.0))      # torchvision F.center_crop


def min_max(x, mean0=True):
    lo = x.min(axis=None, keepdims=True)
    hi = x.max(axis=None, keepdims=True)
    r = (x - lo) / (hi - lo + 1e-8)
    return r * 2 - 1 if mean0 else r


def to_tensor(img_u8_hwc):
    import torch
    return torch.from_numpy(np.ascontiguousarray(img_u8_hwc)).permute(2, 0, 1).contiguous().to(torch.float32).div(255).numpy()


def transform_pil(img_u8, flip, crop=(178, 178), size=(128, 128), minmax=True, mean0=True):
    """One image [H, W, 3] uint8 -> float32 [3, size, size] through Pillow."""
    from PIL import Image
    h, w, _ = img_u8.shape
    top, left = center_crop_box(h, w, *crop)
    im = Image.fromarray(np.ascontiguousarray(img_u8), "RGB").crop((left, top, left + crop[1], top + crop[0]))
    im = im.resize((size[1], size[0]), Image.BILINEAR)
    if flip:
        im = im.transpose(Image.FLIP_LEFT_RIGHT)
    x = to_tensor(np.asarray(im))
    return min_max(x, mean0) if minmax else x


def resize_restated(img_u8, bounds_h, coeffs_h, bounds_v, coeffs_v):
    """Pillow's 8-bit two-pass resample of an [H, W, 3] image from precomputed windows / 22-bit coefficients."""
    prec = 32 - 8 - 2
    h, w, _ = img_u8.shape
    ow, oh = len(bounds_h), len(bounds_v)
    src = img_u8.astype(np.int64)
    tmp = np.zeros((h, ow, 3), np.int64)
    for xx in range(ow):
        x0, n = bounds_h[xx]
        acc = (src[:, x0:x0 + n, :] * coeffs_h[xx, :n].astype(np.int64)[None, :, None]).sum(axis=1) + (1 << (prec - 1))
        tmp[:, xx, :] = np.clip(acc >> prec, 0, 255)
    out = np.zeros((oh, ow, 3), np.int64)
    for yy in range(oh):
        y0, n = bounds_v[yy]
        acc = (tmp[y0:y0 + n] * coeffs_v[yy, :n].astype(np.int64)[:, None, None]).sum(axis=0) + (1 << (prec - 1))
        out[yy] = np.clip(acc >> prec, 0, 255)
    return out.astype(np.uint8)

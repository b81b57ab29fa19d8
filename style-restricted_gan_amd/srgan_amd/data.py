"""GPU side of the reference's input pipeline.

The training notebooks build (05-train cell 9)::

    transforms.Compose([transforms.CenterCrop((178, 178)), transforms.Resize((128, 128)),
                        transforms.RandomHorizontalFlip(p=0.5), transforms.ToTensor(), MinMax(True)])

and apply it per image inside ``FaceDataset.__getitem__`` (pyfiles/dataset.py:127-140) on the CPU.  At several hundred
images per second per GPU that PIL path is the bottleneck, so here the host only DECODES (``FaceDataset`` yields uint8
HWC arrays) and everything after the decode runs in one HIP call per batch on a side stream, from pinned staging buffers:
``GpuTransform`` = crop + Pillow-exact antialiased bilinear resize + flip + ToTensor + MinMax.

The resize tables are built exactly like Pillow's ``precompute_coeffs`` / ``normalize_coeffs_8bpc`` (libImaging/Resample.c,
BILINEAR, support 1.0): float64 windows and weights, normalised, rounded to 22-bit fixed point.  The kernels repeat Pillow's
integer arithmetic, so the resized bytes -- and with IEEE fp32 division the final tensor -- are bit-identical to the
reference transform (tests/test_data_gpu.py checks against PIL itself).
"""
import math

import numpy as np
import torch

from . import _lib
from . import ops

__all__ = ["pil_bilinear_tables", "center_crop_box", "GpuTransform", "PrefetchLoader", "FaceDataset", "get_class_label",
           "pickle_load"]

_PRECISION_BITS = 32 - 8 - 2


def pil_bilinear_tables(in_size, out_size):
    """(bounds [out,2] int32, coeffs [out,ksize] int32, ksize) of one Pillow BILINEAR resample pass over a full axis."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    coeffs = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = np.zeros(ksize, np.float64)
        ww = 0.0
        for x in range(xmax):
            a = abs((x + xmin - center + 0.5) * ss)
            w[x] = 1.0 - a if a < 1.0 else 0.0
            ww += w[x]
        if ww != 0.0:
            w[:xmax] /= ww
        for x in range(ksize):
            v = w[x] * (1 << _PRECISION_BITS)
            coeffs[xx, x] = int(0.5 + v) if w[x] >= 0 else int(-0.5 + v)
        bounds[xx] = (xmin, xmax)
    return bounds, coeffs, ksize


def center_crop_box(h, w, crop_h, crop_w):
    """(top, left) of torchvision.transforms.CenterCrop on an h x w image (no padding case)."""
    if crop_h > h or crop_w > w:
        raise ValueError("GpuTransform: crop larger than the image (torchvision would pad)")
    return int(round((h - crop_h) / 2.0)), int(round((w - crop_w) / 2.0))


class GpuTransform:
    """CenterCrop(crop) -> Resize(size) -> RandomHorizontalFlip(p) -> ToTensor -> MinMax(mean0) on a uint8 batch.

    ``__call__(batch_u8)``: ``batch_u8`` is a uint8 tensor / array [B, H, W, 3] (decoded RGB, host or device).  Returns the
    float32 device batch as a logical [B, 3, size, size] tensor in channels-last memory (what the HIP modules consume).
    Flip decisions are drawn like RandomHorizontalFlip -- one ``torch.rand(1) < p`` per image, in order, from the CPU
    default generator -- unless ``flips`` is given."""

    def __init__(self, crop=(178, 178), size=(128, 128), p=0.5, minmax=True, mean0=True, device="cuda"):
        self.crop, self.size, self.p, self.minmax, self.mean0 = tuple(crop), tuple(size), float(p), bool(minmax), bool(mean0)
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        hb, hc, self.h_ksize = pil_bilinear_tables(self.crop[1], self.size[1])
        vb, vc, self.v_ksize = pil_bilinear_tables(self.crop[0], self.size[0])
        dev = self.device
        self._tables = [torch.from_numpy(a).to(dev).contiguous() for a in (hb, hc, vb, vc)]

    def draw_flips(self, n):
        return torch.tensor([bool(torch.rand(1) < self.p) for _ in range(n)], dtype=torch.uint8)

    def __call__(self, batch_u8, flips=None, stream=None):
        lib = _lib.load()
        x = torch.as_tensor(batch_u8)
        if x.dtype != torch.uint8 or x.dim() != 4 or x.shape[3] != 3:
            raise ValueError("GpuTransform expects a uint8 [B, H, W, 3] batch")
        B, H, W, _ = x.shape
        top, left = center_crop_box(H, W, *self.crop)
        if flips is None:
            flips = self.draw_flips(B)
        flips = torch.as_tensor(flips, dtype=torch.uint8)
        if flips.numel() != B:
            raise ValueError("GpuTransform: one flip flag per image")
        stream = stream or torch.cuda.current_stream(self.device)
        with torch.cuda.stream(stream):
            xd = x.contiguous().to(self.device, non_blocking=True)
            fd = flips.contiguous().to(self.device, non_blocking=True)
            out = torch.empty((B, self.size[0], self.size[1], 3), dtype=torch.float32, device=self.device)
            nbytes = lib.srgan_preprocess_workspace(B, self.crop[0], self.size[0], self.size[1])
            ws = ops.workspace(self.device, nbytes)
            hb, hc, vb, vc = self._tables
            _lib.check(lib.srgan_preprocess_u8(xd.data_ptr(), B, H, W, top, left, self.crop[0], self.crop[1], self.size[0],
                                               self.size[1], hb.data_ptr(), hc.data_ptr(), self.h_ksize, vb.data_ptr(),
                                               vc.data_ptr(), self.v_ksize, fd.data_ptr(), int(self.minmax), int(self.mean0),
                                               out.data_ptr(), ws.data_ptr(), nbytes, stream.cuda_stream), "preprocess_u8")
        # xd / fd / ws were allocated while `stream` was current and are only used on it: the caching allocator already orders
        # their reuse.  `out` is the one tensor that crosses streams -- its CONSUMER must record it (PrefetchLoader does).
        return out.permute(0, 3, 1, 2)        # logical NCHW over NHWC memory


class PrefetchLoader:
    """Wraps an iterable of (uint8 [B,H,W,3] batch, labels): stages the next batch in pinned memory, runs ``transform``
    on a side HIP stream and hands out device batches one step ahead of the consumer (copy and preprocessing of batch
    i+1 overlap the train step of batch i)."""

    def __init__(self, loader, transform):
        self.loader, self.transform = loader, transform
        self.stream = torch.cuda.Stream(device=transform.device)

    def __iter__(self):
        it = iter(self.loader)
        nxt = self._stage(it)
        while nxt is not None:
            cur = nxt
            nxt = self._stage(it)
            consumer = torch.cuda.current_stream(self.transform.device)
            consumer.wait_stream(self.stream)
            # the batch was allocated on the side stream's pool but is read by the train step on the consumer stream: without
            # this its block could be handed to a later _stage() (H2D + preprocess on the side stream, which never waits for
            # the consumer) while the step that reads it is still queued
            cur[0].record_stream(consumer)
            yield cur

    def _stage(self, it):
        try:
            images, labels = next(it)
        except StopIteration:
            return None
        images = torch.as_tensor(images)
        if not images.is_pinned():
            images = images.contiguous().pin_memory()
        return self.transform(images, stream=self.stream), labels


def pickle_load(path):
    """util.pickle_load (pyfiles/util.py:84-106)."""
    import pickle
    with open(path, mode="rb") as f:
        return pickle.load(f)


def get_class_label(n_class_type):
    """Every +-1 sign pattern of length n, in descending order (pyfiles/dataset.py:11-18): row i is domain i."""
    import itertools
    return sorted(itertools.product((1, -1), repeat=n_class_type), reverse=True)


class FaceDataset(torch.utils.data.Dataset):
    """CelebA file selection with the reference's constructor signature and semantics (pyfiles/dataset.py:20-142).

    Label files under ``label_root`` are pickled string arrays ``[n, 1 + attributes]`` (column 0: file name).  An image
    belongs to domain i when its ``dataset_label["class"]`` columns spell sign pattern i, all ``"delete"`` columns are
    "-1" and all ``"existed"`` columns are "1".  Per domain the sorted file list is split into train / val / test
    (``train = first min(train_num, n - val_num - test_num)``, ``val`` the next ``val_num``, ``test`` the last ``test_num``).

    ``__getitem__`` decodes with PIL like the reference.  With a per-sample ``transform`` it behaves exactly as the
    reference; with ``transform=None`` or a :class:`GpuTransform` it returns the decoded image as a uint8 [H, W, 3] tensor,
    to be batched and pushed through ``PrefetchLoader`` (the transform then runs once per batch on the GPU)."""

    def __init__(self, root, label_root, transform, dataset_label, classes, data_type="train", train_num=2000, val_num=500,
                 test_num=500):
        import glob
        self.transform = transform
        self.images, self.labels = [], []
        patterns = get_class_label(len(dataset_label["class"]))
        tables = [np.asarray(pickle_load(path)) for path in sorted(glob.glob(label_root + "*"))]
        per_domain = {}
        for i in range(len(classes)):
            names = []
            for info in tables:
                keep = np.ones(info.shape[0], bool)
                for col in dataset_label["delete"]:
                    keep &= info[:, col] == "-1"
                for col in dataset_label["existed"]:
                    keep &= info[:, col] == "1"
                for col, sign in zip(dataset_label["class"], patterns[i]):
                    keep &= info[:, col] == str(sign)
                names += [root + str(n).split(".")[0] + ".png" for n in info[keep, 0]]
            names.sort()
            n_train = min(train_num, len(names) - val_num - test_num)
            if data_type == "train":
                names = names[:n_train]
            elif data_type == "val":
                names = names[n_train:n_train + val_num]
            elif data_type == "test":
                names = names[-test_num:]
            per_domain[i] = names
        for label in classes:
            self.images += per_domain[label]
            self.labels += [label] * len(per_domain[label])

    def __getitem__(self, index):
        from PIL import Image
        with open(self.images[index], "rb") as f:
            image = Image.open(f).convert("RGB")
        label = self.labels[index]
        if self.transform is None or isinstance(self.transform, GpuTransform):
            return torch.from_numpy(np.array(image, dtype=np.uint8)), label
        return self.transform(image), label

    def __len__(self):
        return len(self.images)

"""Evaluation path of the reference (``pyfiles/evaluation.py``) on the MI355X kernels.

Same names and call signatures as the reference module: ``vgg_model`` (evaluation.py:13-36), ``GAN_evaluation`` with
``preprocess`` / ``get_feature`` / ``get_prdc`` (:38-110), ``evaluation_init`` (:112-122).  Two third-party pieces the reference
imports are not in this image and are provided here:

* ``VGG19_bn`` -- torchvision's ``models.vgg19_bn`` as a parameter holder with torchvision's ``state_dict`` keys, shapes and
  default initialisation, so the reference's ``.pth`` files (ImageNet / the CelebA facial recogniser) load unchanged.  In
  eval mode every BatchNorm is folded into its convolution once (``fold()``); the forward is 16 fused conv + bias + ReLU
  launches on the train step's Winograd / implicit-GEMM kernels, five 2x2 max pools and the two Linear layers as 1x1
  convolutions (flattened in NCHW order like ``torch.flatten``).  Inference only (no backward).
* ``compute_prdc`` -- prdc 0.2's function of that name over the HIP distance / k-th value / set-statistics kernels.
"""
import ctypes

import numpy as np
import torch
import torch.nn as nn

from . import _lib, ops
from .inference import cuda2numpy, image_from_output
from .losses import weights_init
from .ops import ACT_NONE, ACT_RELU, PAD_ZERO

__all__ = ["VGG19_bn", "vgg19_bn", "vgg_model", "GAN_evaluation", "evaluation_init", "compute_prdc", "maxpool2"]

VGG19_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]


def maxpool2(x):
    """nn.MaxPool2d(2, 2) on an NHWC-dense activation (inference)."""
    x = ops.to_nhwc(x)
    n, c, h, w = x.shape
    y = ops.nhwc_empty(n, c, h // 2, w // 2, x.device)
    _lib.check(_lib.load().srgan_maxpool2_fwd(ops._ptr(x), ops._ptr(y), n, h, w, c, ops._stream()), "maxpool2_fwd")
    return y


class _MaxPool(nn.Module):
    def forward(self, x):
        return maxpool2(x)


class _Holder(nn.Module):
    """Parameter-free slot (ReLU / Dropout / AdaptiveAvgPool2d) that keeps torchvision's Sequential indices."""

    def forward(self, x):
        return x


class VGG19_bn(nn.Module):
    def __init__(self, num_classes=1000, width_div=1):
        super().__init__()
        layers, cin = [], 3
        for v in VGG19_CFG:
            if v == "M":
                layers.append(_MaxPool())
                continue
            c = v // width_div
            layers += [nn.Conv2d(cin, c, kernel_size=3, padding=1), nn.BatchNorm2d(c), _Holder()]
            cin = c
        self.features = nn.Sequential(*layers)
        self.avgpool = _Holder()                   # AdaptiveAvgPool2d((7, 7)): the identity on the 224x224 inputs of the path
        hid = 4096 // width_div
        self.classifier = nn.Sequential(nn.Linear(cin * 49, hid), _Holder(), _Holder(), nn.Linear(hid, hid), _Holder(), _Holder(),
                                        nn.Linear(hid, num_classes))
        for m in self.modules():                   # torchvision.models.vgg.VGG._initialize_weights
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, 0, 0.01)
                nn.init.constant_(m.bias, 0)
        self._folded = None

    def fold(self):
        """Eval-mode BatchNorm folded into the preceding convolution: w' = w * g / sqrt(var + eps), b' = (b - mean) * g /
        sqrt(var + eps) + beta.  One-time parameter preparation (re-run after loading weights)."""
        folded, mods = [], list(self.features)
        with torch.no_grad():
            for i, m in enumerate(mods):
                if isinstance(m, nn.Conv2d):
                    bn = mods[i + 1]
                    s = bn.weight / torch.sqrt(bn.running_var + bn.eps)
                    folded.append(((m.weight * s.view(-1, 1, 1, 1)).contiguous(), ((m.bias - bn.running_mean) * s + bn.bias).contiguous()))
                elif isinstance(m, _MaxPool):
                    folded.append(None)
        self._folded = folded
        return self

    def load_state_dict(self, *a, **k):
        self._folded = None
        return super().load_state_dict(*a, **k)

    def _features(self, x):
        if self.training:
            raise NotImplementedError("VGG19_bn runs in eval mode only (the evaluation path never trains it): call .eval()")
        if self._folded is None or self._folded[0][0].device != x.device:
            self.fold()
        x = ops.to_nhwc(x)
        for f in self._folded:
            x = maxpool2(x) if f is None else ops.conv2d(x, f[0], f[1], 1, 1, PAD_ZERO, ACT_RELU)
        n, c, h, w = x.shape
        if (h, w) != (7, 7):
            raise NotImplementedError(f"VGG19_bn: expected 224x224 inputs (7x7 final map), got a {h}x{w} final map")
        return ops.to_nchw(x).reshape(n, c * h * w, 1, 1)          # torch.flatten(x, 1) order

    @staticmethod
    def _linear(x, lin, act):
        return ops.conv2d(x, lin.weight.view(lin.out_features, lin.in_features, 1, 1), lin.bias, 1, 0, PAD_ZERO, act)

    def feature(self, x):
        """features -> avgpool -> flatten -> classifier[:6]  ([N, 4096])."""
        with torch.no_grad(), ops.pack_cache():
            h = self._features(x)
            h = self._linear(h, self.classifier[0], ACT_RELU)
            h = self._linear(h, self.classifier[3], ACT_RELU)
            return h.reshape(h.shape[0], -1)

    def forward(self, x):
        with torch.no_grad(), ops.pack_cache():
            h = self.feature(x)
            h = self._linear(h.reshape(h.shape[0], -1, 1, 1), self.classifier[6], ACT_NONE)
            return h.reshape(h.shape[0], -1)


def vgg19_bn(pretrained=False, **kw):
    """``torchvision.models.vgg19_bn``.  ``pretrained=True`` needs torchvision's weight file, which this image does not have:
    load it yourself with ``load_state_dict`` (the keys are torchvision's)."""
    if pretrained:
        raise NotImplementedError("vgg19_bn(pretrained=True): no network / torchvision here -- build the model and "
                                  "load_state_dict() the ImageNet weights from a file")
    return VGG19_bn(**kw)


class vgg_model():
    """evaluation.py:13-36: ``get(x, "feature")`` = classifier[:6] output, ``get(x, "score")`` = the full model."""

    def __init__(self, model):
        self.model = model

    def get(self, x, output_type="score"):
        if output_type == "feature":
            return self.model.feature(x)
        elif output_type == "score":
            return self.model(x)


def _dev_f32(a, device):
    t = torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32) if not torch.is_tensor(a) else a.to(torch.float32)
    return t.to(device).contiguous()


def compute_prdc(real_features, fake_features, nearest_k, device="cuda"):
    """prdc.compute_prdc(real_features, fake_features, nearest_k) -> dict(precision, recall, density, coverage).
    ``nearest_k`` <= 15 (the reference's notebooks use 5, evaluation.py:85-110): the k-th-neighbour kernel keeps its candidates
    in registers (``KTH_MAX`` = 16 slots per lane, csrc/evaluation.hip); a larger k raises ``ValueError`` here instead of a C-ABI
    error from inside the launch sequence."""
    if not 1 <= int(nearest_k) <= 15:
        raise ValueError(f"compute_prdc: nearest_k = {nearest_k} outside 1..15 (register-resident k-th-neighbour kernel)")
    lib = _lib.load()
    x, y = _dev_f32(real_features, device), _dev_f32(fake_features, device)
    n, d = x.shape
    m = y.shape[0]
    if y.shape[1] != d:
        raise ValueError("compute_prdc: feature dimensions differ")
    st = ops._stream()

    def dist(a, b):
        out = torch.empty(a.shape[0], b.shape[0], dtype=torch.float32, device=a.device)
        _lib.check(lib.srgan_pairwise_dist(ops._ptr(a), a.shape[0], ops._ptr(b), b.shape[0], d, ops._ptr(out), st), "pairwise_dist")
        return out

    def radii(a):
        dm = dist(a, a)
        out = torch.empty(a.shape[0], dtype=torch.float32, device=a.device)
        _lib.check(lib.srgan_kth_smallest_rows(ops._ptr(dm), a.shape[0], a.shape[0], nearest_k + 1, ops._ptr(out), st), "kth_smallest_rows")
        return out

    r_real, r_fake = radii(x), radii(y)
    drf = dist(x, y)
    out4 = torch.empty(4, dtype=torch.float32, device=x.device)
    nb = lib.srgan_prdc_workspace(n, m)
    ws = torch.empty(nb, dtype=torch.uint8, device=x.device)
    _lib.check(lib.srgan_prdc_from_dist(ops._ptr(drf), n, m, ops._ptr(r_real), ops._ptr(r_fake), int(nearest_k), ops._ptr(out4),
                                        ops._ptr(ws), nb, st), "prdc_from_dist")
    p, r, dn, c = (float(v) for v in out4.cpu())
    return dict(precision=p, recall=r, density=dn, coverage=c)


class GAN_evaluation():
    """evaluation.py:38-110.  ``feature_extractor``: "vgg-initialization" (default-initialised VGG19-bn, as the reference's
    ``models.vgg19_bn(pretrained=False)`` + the no-op ``weights_init``), or "vgg-ImageNet" / "vgg-CelebA" with ``weights=`` a
    path to (or a dict of) torchvision-keyed parameters -- the reference downloads / reads them from files this image lacks."""

    def __init__(self, feature_extractor="vgg-initialization", device="cuda", classes=tuple(range(4)), reference=tuple(range(4)),
                 weights=None):
        self.fe = feature_extractor
        if "vgg" not in self.fe:
            raise NotImplementedError("GAN_evaluation: only the vgg feature extractors of the reference exist")
        if "CelebA" in self.fe:
            model = VGG19_bn(num_classes=len(classes))
        else:
            model = VGG19_bn()
        if "initialization" in self.fe:
            model.apply(weights_init)
        else:
            if weights is None:
                raise FileNotFoundError(f"GAN_evaluation({self.fe!r}): pass weights=<.pth path or state_dict> (torchvision keys); "
                                        "the reference's files are not in this image")
            sd = torch.load(weights, map_location="cpu") if isinstance(weights, str) else weights
            model.load_state_dict(sd)
        model.eval()
        self.model = vgg_model(model.to(device))
        self.device = device

    def preprocess(self, tensor):
        """Every image: 8-bit rendering (``image_from_output``), Resize((128,128)) -> Resize((224,224)) -> ToTensor ->
        Normalize(ImageNet mean / std)   (evaluation.py:61-66, 72-81).  Host side, as in the reference (PIL)."""
        from PIL import Image
        mean = np.asarray((0.485, 0.456, 0.406), np.float32)[:, None, None]
        std = np.asarray((0.229, 0.224, 0.225), np.float32)[:, None, None]
        images = []
        for i in range(tensor.shape[0]):
            im = image_from_output(tensor[i:i + 1, :, :, :])[0]
            im = im.resize((128, 128), Image.BILINEAR).resize((224, 224), Image.BILINEAR)
            a = np.asarray(im, dtype=np.float32).transpose(2, 0, 1) / 255.0
            images.append((a - mean) / std)
        return torch.Tensor(np.array(images))

    def get_feature(self, tensor, batch=32, get_attention=False, thres=0.5):
        num = tensor.shape[0]
        features = None
        for itr in range(num // batch + int(bool(num - batch * (num // batch)))):
            data = tensor[itr * batch:(itr + 1) * batch].to(self.device)
            feature = cuda2numpy(self.model.get(data, "feature").reshape(data.shape[0], -1))
            features = feature if features is None else np.concatenate([features, feature], axis=0)
        return features

    def get_prdc(self, true, pred, nearest_k=5, preprocess=True, thres=0.5, batch=32):
        self.run_preprocess = preprocess
        if preprocess:
            true = self.preprocess(true)
            pred = self.preprocess(pred)
        f1 = self.get_feature(true)
        f2 = self.get_feature(pred)
        if f1.shape[1] == 0:
            return {"precision": None, "recall": None, "density": None, "coverage": None}
        return compute_prdc(real_features=f1, fake_features=f2, nearest_k=nearest_k, device=self.device)


def evaluation_init(fe_list, classes, metrics):
    """Nested result store fe -> source -> target -> metric -> []   (evaluation.py:112-122)."""
    return {fe: {s: {t: {m: [] for m in metrics.keys()} for t in classes} for s in classes} for fe in fe_list}

"""CPU restatement of the reference's evaluation path: VGG19-bn feature extractor + PRDC.

TEST INFRASTRUCTURE (see oracle/__init__.py).

Reference sites restated here:
* ``vgg19_bn_spec`` / ``vgg19_bn_features``  <- ``vgg_model.get(x, "feature")``      pyfiles/evaluation.py:13-36
      = ``torchvision.models.vgg19_bn`` (.features, .avgpool, .classifier[:6]) in eval mode.
* ``preprocess``                             <- ``GAN_evaluation.preprocess``        pyfiles/evaluation.py:72-81 (+ :61-66)
* ``compute_prdc``                           <- ``prdc.compute_prdc``                pyfiles/evaluation.py:98-110

PARITY UNPINNED for this file.  Both third-party pieces are absent from /root/reference and from this image:
``torchvision==0.5.0`` (Docker/requirements.txt:9) and ``prdc==0.2`` (Docker/requirements.txt:13), the reference holds no
fixture for either, and its weight file (data/parameters/B/facial_recognizer_vgg_*.pth) is a git-LFS pointer.  What is
restated is their PUBLISHED definition:
  * torchvision's VGG configuration "E" with batch norm: [64, 64, M, 128, 128, M, 256 x4, M, 512 x4, M, 512 x4, M], every conv
    3x3 / stride 1 / pad 1 with bias followed by BatchNorm2d and ReLU, M = MaxPool2d(2, 2), AdaptiveAvgPool2d((7, 7)),
    classifier = Linear(25088, 4096) ReLU Dropout Linear(4096, 4096) ReLU Dropout Linear(4096, classes); state_dict keys
    ``features.<i>.*`` / ``classifier.<i>.*``; default init kaiming_normal_(fan_out, relu) / BN 1, 0 / Linear N(0, 0.01);
  * prdc 0.2 (Naeem, Oh, Uh, Choi, Yoo: "Reliable Fidelity and Diversity Metrics for Generative Models", ICML 2020):
    Euclidean pairwise distances (sklearn.metrics.pairwise_distances), radius of a sample = its (k+1)-th smallest distance
    to its own set (itself included), then the four set statistics below.
The arithmetic itself is held to independent implementations that ARE installed here: torch's own conv / batch_norm /
max_pool / linear ops and scikit-learn's ``pairwise_distances`` (tests/test_evaluation_cpu.py).
"""
import numpy as np
import torch
import torch.nn.functional as F

VGG19_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]
BN_EPS = 1e-5
IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def vgg19_bn_spec(num_classes=1000, width_div=1):
    """{state_dict key: shape} in torchvision's order.  ``width_div`` > 1 shrinks every width (tests only)."""
    spec, i, cin = {}, 0, 3
    for v in VGG19_CFG:
        if v == "M":
            i += 1
            continue
        c = v // width_div
        spec[f"features.{i}.weight"], spec[f"features.{i}.bias"] = (c, cin, 3, 3), (c,)
        for k in ("weight", "bias", "running_mean", "running_var"):
            spec[f"features.{i + 1}.{k}"] = (c,)
        spec[f"features.{i + 1}.num_batches_tracked"] = ()
        cin, i = c, i + 3
    hid = 4096 // width_div
    spec["classifier.0.weight"], spec["classifier.0.bias"] = (hid, cin * 49), (hid,)
    spec["classifier.3.weight"], spec["classifier.3.bias"] = (hid, hid), (hid,)
    spec["classifier.6.weight"], spec["classifier.6.bias"] = (num_classes, hid), (num_classes,)
    return spec


def fill(spec, seed=0):
    """Deterministic parameters with trained-looking statistics (BN running stats away from 0 / 1)."""
    g = torch.Generator().manual_seed(seed)
    P = {}
    for k, shape in spec.items():
        if k.endswith("num_batches_tracked"):
            P[k] = torch.tensor(100, dtype=torch.int64)
        elif k.endswith("running_var"):
            P[k] = torch.rand(shape, generator=g) * 0.5 + 0.75
        elif k.endswith("running_mean"):
            P[k] = torch.randn(shape, generator=g) * 0.1
        elif len(shape) == 1 and "features" in k and int(k.split(".")[1]) % 3 != 0 and k.endswith("weight"):
            P[k] = torch.rand(shape, generator=g) * 0.5 + 0.75              # BN gamma
        elif len(shape) == 1:
            P[k] = torch.randn(shape, generator=g) * 0.05
        else:
            fan_in = int(np.prod(shape[1:]))
            P[k] = torch.randn(shape, generator=g) * (2.0 / fan_in) ** 0.5
    return P


def vgg19_bn_features(P, x):
    """``vgg_model.get(x, "feature")``: features -> avgpool(7x7) -> flatten (NCHW order) -> classifier[:6], eval mode."""
    i = 0
    for v in VGG19_CFG:
        if v == "M":
            x = F.max_pool2d(x, 2, 2)
            i += 1
            continue
        x = F.conv2d(x, P[f"features.{i}.weight"], P[f"features.{i}.bias"], 1, 1)
        x = F.batch_norm(x, P[f"features.{i + 1}.running_mean"], P[f"features.{i + 1}.running_var"],
                         P[f"features.{i + 1}.weight"], P[f"features.{i + 1}.bias"], False, 0.0, BN_EPS)
        x = torch.relu(x)
        i += 3
    x = F.adaptive_avg_pool2d(x, (7, 7))
    x = torch.flatten(x, 1)
    x = torch.relu(F.linear(x, P["classifier.0.weight"], P["classifier.0.bias"]))       # Dropout: identity in eval mode
    return torch.relu(F.linear(x, P["classifier.3.weight"], P["classifier.3.bias"]))


def vgg19_bn_scores(P, x):
    return F.linear(vgg19_bn_features(P, x), P["classifier.6.weight"], P["classifier.6.bias"])


def image_from_output_u8(t):
    """One [c, h, w] tensor -> uint8 [h, w, 3] as util.image_from_output (pyfiles/util.py:157-188)."""
    a = t.detach().cpu().numpy()
    a = np.tile(np.transpose(a, (1, 2, 0)), (1, 1, int(3 / a.shape[0])))
    lo, hi = a.min(axis=None, keepdims=True), a.max(axis=None, keepdims=True)
    a = (a - lo) / (hi - lo + 1e-8) * 2 ** 8
    a[a > 255] = 255
    return np.uint8(a)


def preprocess(tensor, size=224):
    """Resize((128,128)) -> Resize((224,224)) -> ToTensor -> Normalize(ImageNet) of the 8-bit rendering of every image."""
    from PIL import Image
    out = []
    for i in range(tensor.shape[0]):
        im = Image.fromarray(image_from_output_u8(tensor[i]))
        im = im.resize((128, 128), Image.BILINEAR).resize((size, size), Image.BILINEAR)
        a = np.asarray(im, dtype=np.float32).transpose(2, 0, 1) / 255.0
        a = (a - np.asarray(IMAGENET_MEAN, np.float32)[:, None, None]) / np.asarray(IMAGENET_STD, np.float32)[:, None, None]
        out.append(a)
    return torch.tensor(np.array(out))


# ---- prdc 0.2 -------------------------------------------------------------------------------------------------------
def pairwise_distance(x, y=None):
    x = np.asarray(x, dtype=np.float64)
    y = x if y is None else np.asarray(y, dtype=np.float64)
    d2 = ((x[:, None, :] - y[None, :, :]) ** 2).sum(-1) if x.shape[0] * y.shape[0] * x.shape[1] <= 2 ** 26 else None
    if d2 is None:
        d2 = np.maximum((x * x).sum(1)[:, None] + (y * y).sum(1)[None, :] - 2.0 * x @ y.T, 0.0)
    return np.sqrt(d2)


def kth_value(unsorted, k, axis=-1):
    idx = np.argpartition(unsorted, k - 1, axis=axis)[..., :k]
    return np.take_along_axis(unsorted, idx, axis=axis).max(axis=axis)


def nearest_neighbour_distances(features, nearest_k):
    return kth_value(pairwise_distance(features), nearest_k + 1, axis=-1)


def compute_prdc(real_features, fake_features, nearest_k):
    r_real = nearest_neighbour_distances(real_features, nearest_k)
    r_fake = nearest_neighbour_distances(fake_features, nearest_k)
    d = pairwise_distance(real_features, fake_features)
    inside_real = d < r_real[:, None]
    precision = inside_real.any(axis=0).mean()
    recall = (d < r_fake[None, :]).any(axis=1).mean()
    density = (1.0 / float(nearest_k)) * inside_real.sum(axis=0).mean()
    coverage = (d.min(axis=1) < r_real).mean()
    return dict(precision=float(precision), recall=float(recall), density=float(density), coverage=float(coverage))

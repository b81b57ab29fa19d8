"""Parameter shape specs + the build-owned deterministic fill (oracle side).

TEST INFRASTRUCTURE (see oracle/__init__.py).

Shape specs restate the constructors of the reference networks so that the
oracle can run on a flat ``{state_dict key: tensor}`` dictionary:

* ``generator_spec``      <- SingleGenerator.__init__        (pyfiles/model.py:204-234)
* ``discriminator_spec``  <- SingleDiscriminator_solo_multi  (pyfiles/model.py:294-337)
* ``discriminator_original_spec`` <- SingleDiscriminator_original_multi (model.py:255-287)
* ``encoder_spec``        <- Encoder.__init__                (pyfiles/model.py:440-457)
* ``encoder_original_spec`` <- Encoder_original.__init__     (pyfiles/model.py:379-396)

Key order equals the reference ``state_dict()`` order;
``tests/golden/shapes_*.json`` (written from the imported reference) pins it.

``fill`` is NOT the reference's initialiser (the reference keeps PyTorch's
default init because ``weights_init`` never matches, util.py:193-203).  It is a
deterministic, platform-independent fill both the oracle and the HIP path can
regenerate from ``(seed, key)`` so that goldens need not store 90 MB of weights.
"""
from collections import OrderedDict
import zlib

import numpy as np
import torch


def _cbin(spec, prefix, ch, num_con):
    spec[prefix + ".weight"] = (ch,)
    spec[prefix + ".bias"] = (ch,)
    spec[prefix + ".ConBias.0.weight"] = (ch, num_con)
    spec[prefix + ".ConBias.0.bias"] = (ch,)


def generator_spec(nch_in, nch, reduce=2, num_cls=3, res_num=6, num_con=2, nch_out=None):
    nch_out = nch_in if nch_out is None else nch_out
    k = 2 * reduce
    s = OrderedDict()
    s["down_convs.0.weight"] = (nch, nch_in, 7, 7)
    for i in range(num_cls):
        s[f"down_convs.{i + 1}.weight"] = (nch * 2 ** (i + 1), nch * 2 ** i, k, k)
    for i in range(num_cls + 1):
        _cbin(s, f"down_cnorms.{i}", nch * 2 ** i, num_con)
    top = nch * 2 ** num_cls
    for j in range(res_num):
        s[f"resBlocks.{j}.c1.weight"] = (top, top, 3, 3)
        _cbin(s, f"resBlocks.{j}.cn1", top, num_con)
        s[f"resBlocks.{j}.c2.weight"] = (top, top, 3, 3)
        _cbin(s, f"resBlocks.{j}.cn2", top, num_con)
    # transposed convs store [Cin, Cout, kh, kw]
    idx = 0
    for i in range(num_cls, 0, -1):
        s[f"up_convs.{idx}.weight"] = (nch * 2 ** i, nch * 2 ** (i - 1), k, k)
        idx += 1
    s[f"up_convs.{idx}.weight"] = (nch_out, nch, 7, 7)
    return s


def _trunk_widths(nch, num_cls):
    widths = [nch]
    for _ in range(1, num_cls):
        widths.append(min(widths[-1] * 2, nch * 8))
    return widths


def discriminator_spec(nch_in, nch, reduce=2, num_cls=3, n_class=4):
    k = 2 * reduce
    s = OrderedDict()
    for d, width in (("discriminator1", nch), ("discriminator2", nch // 2)):
        w = _trunk_widths(width, num_cls)
        s[f"{d}.down_convs.0.weight"] = (w[0], nch_in, 4, 4)
        for i in range(1, num_cls):
            s[f"{d}.down_convs.{2 * i}.weight"] = (w[i], w[i - 1], k, k)
    dim_in = min(nch * 2 ** num_cls, nch * 8)
    s["last_layer1.weight"] = (1, dim_in, 4, 4)
    s["last_layer1.bias"] = (1,)
    s["last_layer2.weight"] = (1, dim_in // 2, 4, 4)
    s["last_layer2.bias"] = (1,)
    s["classification_layer1.0.weight"] = (n_class, dim_in, 8, 8)
    s["classification_layer1.0.bias"] = (n_class,)
    s["classification_layer2.0.weight"] = (n_class, dim_in // 2, 4, 4)
    s["classification_layer2.0.bias"] = (n_class,)
    return s


def discriminator_original_spec(nch_in, nch, reduce=2, num_cls=3):
    k = 2 * reduce
    s = OrderedDict()
    for d, width in (("discriminator1", nch), ("discriminator2", nch // 2)):
        w = _trunk_widths(width, num_cls)
        s[f"{d}.down_convs.0.weight"] = (w[0], nch_in, 4, 4)
        for i in range(1, num_cls):
            s[f"{d}.down_convs.{2 * i}.weight"] = (w[i], w[i - 1], k, k)
        s[f"{d}.down_convs.{2 * num_cls}.weight"] = (1, w[-1], 4, 4)
        s[f"{d}.down_convs.{2 * num_cls}.bias"] = (1,)
    return s


def encoder_spec(nch_in, nch_out, nch=64, num_cls=3, num_con=2):
    s = OrderedDict()
    s["first_layer.weight"] = (nch, nch_in, 7, 7)
    s["first_layer.bias"] = (nch,)
    c = nch
    for b in range(num_cls):
        s[f"layers.{b}.conv1.weight"] = (c, c, 3, 3)
        s[f"layers.{b}.cmp.0.weight"] = (2 * c, c, 3, 3)
        s[f"layers.{b}.shortcut.1.weight"] = (2 * c, c, 1, 1)
        s[f"layers.{b}.shortcut.1.bias"] = (2 * c,)
        c *= 2
    for head, n in (("fcmean", nch_out), ("fcvar", nch_out), ("fcclass", num_con)):
        s[f"{head}.weight"] = (n, c)
        s[f"{head}.bias"] = (n,)
    return s


def encoder_original_spec(nch_in, nch_out, nch=64, num_cls=3, num_con=2):
    s = OrderedDict()
    s["first_layer.weight"] = (nch, nch_in, 7, 7)
    s["first_layer.bias"] = (nch,)
    c = nch
    for b in range(num_cls):
        _cbin(s, f"layers.{b}.cnorm1", c, num_con)
        s[f"layers.{b}.conv1.weight"] = (c, c, 3, 3)
        _cbin(s, f"layers.{b}.cnorm2", c, num_con)
        s[f"layers.{b}.cmp.0.weight"] = (2 * c, c, 3, 3)
        s[f"layers.{b}.shortcut.1.weight"] = (2 * c, c, 1, 1)
        s[f"layers.{b}.shortcut.1.bias"] = (2 * c,)
        c *= 2
    for head in ("fcmean", "fcvar"):
        s[f"{head}.weight"] = (nch_out, c)
        s[f"{head}.bias"] = (nch_out,)
    return s


def fill_array(key, shape, seed=0):
    """Deterministic value for one tensor: PCG64 seeded by (seed, crc32(key))."""
    rng = np.random.Generator(np.random.PCG64([int(seed), zlib.crc32(key.encode())]))
    u = rng.random(size=shape, dtype=np.float64) * 2.0 - 1.0
    is_norm_affine = key.endswith(("cn1.weight", "cn2.weight", "cnorm1.weight", "cnorm2.weight")) or (
        "cnorms." in key and key.endswith(".weight") and "ConBias" not in key)
    if is_norm_affine:          # gamma around 1 so the affine path is exercised
        a = 1.0 + 0.25 * u
    elif len(shape) == 1:       # biases / beta
        a = 0.1 * u
    else:                        # fan-in scaled weights (He-uniform-like gain)
        fan_in = int(np.prod(shape[1:]))
        a = u * np.sqrt(3.0 / fan_in)
    return a.astype(np.float32)


def fill(spec, seed=0, dtype=torch.float32):
    """{key: tensor} for a whole network, regenerated from (seed, key)."""
    return OrderedDict((k, torch.from_numpy(fill_array(k, shp, seed)).to(dtype)) for k, shp in spec.items())

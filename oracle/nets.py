"""Functional PyTorch-CPU restatement of the reference networks (NCHW).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Every function takes a flat
``{state_dict key: tensor}`` dictionary ``P`` (see oracle/params.py) so the
same numbers can be loaded into the reference, the oracle and the HIP modules.

Reference sites restated here:
* ``cbin``            <- _CBINorm.forward                  pyfiles/model.py:54-67
* ``inorm``           <- nn.InstanceNorm2d(affine=False)    pyfiles/model.py:178
* ``generator``       <- SingleGenerator.forward           pyfiles/model.py:236-249
                         SingleResidualBlock.forward       pyfiles/model.py:196-201
* ``discriminator``   <- SingleDiscriminator_solo_multi    pyfiles/model.py:339-346
* ``discriminator_original`` <- SingleDiscriminator_original_multi  model.py:289-292
* ``encoder``         <- Encoder.forward / reparametrize   pyfiles/model.py:459-482
                         BasicBlock_classification.forward pyfiles/model.py:433-437
* ``encoder_original``<- Encoder_original.forward          pyfiles/model.py:398-411
"""
import torch
import torch.nn.functional as F

EPS = 1e-5  # _BatchNorm default eps used by CBINorm2d / InstanceNorm2d


def _plane_stats(x):
    mean = x.mean(dim=(2, 3), keepdim=True)
    var = ((x - mean) ** 2).mean(dim=(2, 3), keepdim=True)  # biased, as instance_norm
    return mean, var


def inorm(x):
    mean, var = _plane_stats(x)
    return (x - mean) / torch.sqrt(var + EPS)


def cbin(x, c, P, prefix):
    """(IN(x) + tanh(Linear(c))) * gamma + beta ; raises like CBINorm2d on non-4D."""
    if x.dim() != 4:
        raise ValueError('expected 4D input (got {}D input)'.format(x.dim()))
    t = torch.tanh(F.linear(c, P[prefix + ".ConBias.0.weight"], P[prefix + ".ConBias.0.bias"]))
    g = P[prefix + ".weight"].view(1, -1, 1, 1)
    b = P[prefix + ".bias"].view(1, -1, 1, 1)
    return (inorm(x) + t[:, :, None, None]) * g + b


def _count(P, fmt):
    n = 0
    while fmt.format(n) in P:
        n += 1
    return n


def generator(P, x, c, reduce=2):
    n_down = _count(P, "down_convs.{}.weight")          # num_cls + 1
    n_res = _count(P, "resBlocks.{}.c1.weight")
    n_up = _count(P, "up_convs.{}.weight")              # num_cls + 1
    pad = reduce // 2
    for i in range(n_down):
        if i == 0:
            x = F.conv2d(x, P["down_convs.0.weight"], None, 1, 3)
        else:
            x = F.conv2d(x, P[f"down_convs.{i}.weight"], None, reduce, pad)
        x = torch.relu(cbin(x, c, P, f"down_cnorms.{i}"))
    for j in range(n_res):
        h = F.conv2d(x, P[f"resBlocks.{j}.c1.weight"], None, 1, 1)
        h = torch.relu(cbin(h, c, P, f"resBlocks.{j}.cn1"))
        h = F.conv2d(h, P[f"resBlocks.{j}.c2.weight"], None, 1, 1)
        x = cbin(h, c, P, f"resBlocks.{j}.cn2") + x
    for i in range(n_up - 1):
        x = F.conv_transpose2d(x, P[f"up_convs.{i}.weight"], None, reduce, pad)
        x = torch.relu(inorm(x))
    x = F.conv2d(x, P[f"up_convs.{n_up - 1}.weight"], None, 1, 3)
    return torch.tanh(x)


D_SLOPE = 0.01   # nn.LeakyReLU() default, pyfiles/model.py:263,303
E_SLOPE = 0.2    # nn.LeakyReLU(0.2),     pyfiles/model.py:418,454


def _pool3s2(x):
    return F.avg_pool2d(x, 3, stride=2, padding=1, count_include_pad=False)


def _trunk(P, prefix, x, reduce=2):
    """[conv k(2*reduce) s(reduce) -> LeakyReLU(0.01)]*; convs sit at even Sequential indices.
    A conv that carries a bias is the 1-channel head of the *_original variant, not trunk."""
    i = 0
    while True:
        key = f"{prefix}.down_convs.{2 * i}"
        if key + ".weight" not in P or key + ".bias" in P:
            break
        stride, pad = (2, 1) if i == 0 else (reduce, reduce // 2)
        x = F.leaky_relu(F.conv2d(x, P[key + ".weight"], None, stride, pad), D_SLOPE)
        i += 1
    return x, i


def discriminator(P, x, n_class=4, reduce=2):
    """-> ([out1, out2], [cls1, cls2]) ; softmax over dim 1 (nn.Softmax() on a 4-D tensor)."""
    d1, _ = _trunk(P, "discriminator1", x, reduce)
    d2, _ = _trunk(P, "discriminator2", _pool3s2(x), reduce)
    o1 = F.conv2d(d1, P["last_layer1.weight"], P["last_layer1.bias"], 1, 1)
    o2 = F.conv2d(d2, P["last_layer2.weight"], P["last_layer2.bias"], 1, 1)
    c1 = F.conv2d(d1, P["classification_layer1.0.weight"], P["classification_layer1.0.bias"])
    c2 = F.conv2d(d2, P["classification_layer2.0.weight"], P["classification_layer2.0.bias"])
    c1 = torch.softmax(c1, dim=1).reshape(-1, n_class)
    c2 = torch.softmax(c2, dim=1).reshape(-1, n_class)
    return [o1, o2], [c1, c2]


def discriminator_original(P, x, reduce=2):
    outs = []
    for prefix, inp in (("discriminator1", x), ("discriminator2", _pool3s2(x))):
        h, n = _trunk(P, prefix, inp, reduce)
        outs.append(F.conv2d(h, P[f"{prefix}.down_convs.{2 * n}.weight"], P[f"{prefix}.down_convs.{2 * n}.bias"], 1, 1))
    return outs


def _conv3_reflect(x, w):
    return F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), w)


def _enc_block(P, b, x, norm1, norm2):
    h = F.leaky_relu(norm1(x), E_SLOPE)
    h = _conv3_reflect(h, P[f"layers.{b}.conv1.weight"])
    h = F.leaky_relu(norm2(h), E_SLOPE)
    h = F.avg_pool2d(_conv3_reflect(h, P[f"layers.{b}.cmp.0.weight"]), 2, 2)
    s = F.conv2d(F.avg_pool2d(x, 2, 2), P[f"layers.{b}.shortcut.1.weight"], P[f"layers.{b}.shortcut.1.bias"])
    return h + s


def _reparam(mu, logvar, noise):
    # eps * exp(logvar/2) + mu ; eps comes from the CPU default generator in the reference
    if noise is None:
        noise = torch.FloatTensor(mu.size()).normal_().to(mu.dtype)
    return noise * torch.exp(0.5 * logvar) + mu


def encoder(P, x, noise=None):
    """-> (c_code, mu, logvar, class_output, None)."""
    h = F.conv2d(x, P["first_layer.weight"], P["first_layer.bias"], 2, 1)
    for b in range(_count(P, "layers.{}.conv1.weight")):
        h = _enc_block(P, b, h, inorm, inorm)
    feat = F.leaky_relu(h, E_SLOPE).mean(dim=(2, 3))
    mu = F.linear(feat, P["fcmean.weight"], P["fcmean.bias"])
    logvar = F.linear(feat, P["fcvar.weight"], P["fcvar.bias"])
    code = _reparam(mu, logvar, noise)
    cls = F.linear(feat, P["fcclass.weight"], P["fcclass.bias"])
    return code, mu, logvar, cls, None


def encoder_original(P, x, c, noise=None):
    """-> (c_code, mu, logvar) ; CBIN conditioned on the class vector c."""
    h = F.conv2d(x, P["first_layer.weight"], P["first_layer.bias"], 2, 1)
    for b in range(_count(P, "layers.{}.conv1.weight")):
        h = _enc_block(P, b, h,
                       lambda t, b=b: cbin(t, c, P, f"layers.{b}.cnorm1"),
                       lambda t, b=b: cbin(t, c, P, f"layers.{b}.cnorm2"))
    feat = F.leaky_relu(h, E_SLOPE).mean(dim=(2, 3))
    mu = F.linear(feat, P["fcmean.weight"], P["fcmean.bias"])
    logvar = F.linear(feat, P["fcvar.weight"], P["fcvar.bias"])
    return _reparam(mu, logvar, noise), mu, logvar


def encoder_classifier(P, x):
    """Encoder_classifier.forward (pyfiles/model.py:503-507): trunk -> LeakyReLU -> global pool -> fcclass -> softmax."""
    h = F.conv2d(x, P["first_layer.weight"], P["first_layer.bias"], 2, 1)
    for b in range(_count(P, "layers.{}.conv1.weight")):
        h = _enc_block(P, b, h, inorm, inorm)
    feat = F.leaky_relu(h, E_SLOPE).mean(dim=(2, 3))
    return torch.softmax(F.linear(feat, P["fcclass.weight"], P["fcclass.bias"]), dim=1)

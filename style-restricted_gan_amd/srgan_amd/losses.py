"""Loss / helper surface of the reference's ``pyfiles/util.py`` (hot-path part) over HIP kernels.

Same names and argument meaning as util.py:193-319 and util.py:455-553; the arithmetic runs in the
fused reductions of csrc/losses.hip.  Host-only helpers (``class_encode``, ``get_target``,
``weights_init``, ``load_classifier``) are restated as plain host logic.
"""
import numpy as np
import torch
import torch.nn as nn

from . import ops

__all__ = ["CrossEntropyLoss", "get_loss_D", "get_domainloss_D", "corrcoef", "corrcoef_loss", "GaussianHistogram",
           "histogram_imitation", "class_encode", "get_target", "weights_init", "load_classifier"]


class CrossEntropyLoss(nn.Module):
    """nn.CrossEntropyLoss() (mean reduction) on the HIP path: the criterion of the encoder pre-training job
    (04_Facial_Recognition-Encoder.ipynb cell 18; note the notebook applies it to the SOFTMAX output of
    Encoder_classifier, i.e. a second softmax is taken inside the loss -- reproduced as is)."""

    def forward(self, input, target):
        return ops.softmax_xent(input, target, 1.0)


def _require_mse(criterion, who):
    if not isinstance(criterion, nn.MSELoss) or getattr(criterion, "reduction", "mean") != "mean":
        raise NotImplementedError(f"{who}: only nn.MSELoss() (mean reduction) has a HIP kernel -- the criterion "
                                  "every reference notebook passes (05-train cell 13)")


def get_loss_D(outputs, target, criterion, device="cuda"):
    """mean over the scales of criterion(output, full_like(output, target))   (util.py:457-462)."""
    _require_mse(criterion, "get_loss_D")
    loss = 0.0
    w = 1.0 / len(outputs)
    for output in outputs:
        loss = loss + ops.mse_const(output, float(target), w)
    return loss


def get_domainloss_D(outputs_class, true_label, criterion_class):
    """mean over the scales of criterion_class(class probabilities, one-hot label)   (util.py:464-468)."""
    _require_mse(criterion_class, "get_domainloss_D")
    loss = 0.0
    w = 1.0 / len(outputs_class)
    for q in outputs_class:
        loss = loss + ops.mse_pair(q, true_label, w)
    return loss


def _latent(mu, n_batch, target, w):
    if target is None:
        target = torch.full((50,), 1.0 / 50, dtype=torch.float32, device=mu.device)
    return ops.latent_losses(mu, n_batch, target, *w)


def corrcoef(x):
    """Row-wise Pearson matrix of x [d, n] with the +-1 clamp (np.corrcoef semantics, util.py:470-511)."""
    _, _, r = _latent(x.t(), max(x.shape[1], 2), None, (0.0, 0.0, 0.0))
    return r


def corrcoef_loss(m, device=None):
    """sum |corrcoef(m) - I| / (d (d-1))   (util.py:513-517); differentiable w.r.t. m."""
    total, _, _ = _latent(m.t(), max(m.shape[1], 2), None, (0.0, 1.0, 0.0))
    return total


class GaussianHistogram(nn.Module):
    """Differentiable Gaussian-kernel histogram of a 1-D sample (util.py:521-537)."""

    def __init__(self, bins, min, max, sigma):
        super().__init__()
        self.bins, self.min, self.max, self.sigma = bins, min, max, sigma
        self.delta = float(max - min) / float(bins)
        self.centers = float(min) + self.delta * (torch.arange(bins).float() + 0.5)

    def forward(self, x):
        return ops.soft_histogram(x, self.bins, float(self.min), float(self.max), float(self.sigma))


class histogram_imitation():
    """KL(target || soft-histogram(x[:, j])) summed over latent dims (util.py:539-553).

    The target is built once from ``torch.randn(target_num, 1)`` drawn from the CPU default generator --
    like the reference constructor, this advances the global RNG."""

    def __init__(self, device, bins=50, range_max=10, sigma=0.2, target_num=100000):
        self.device = device
        self.bins, self.range_max, self.sigma = bins, float(range_max), float(sigma)
        self.gausshist = GaussianHistogram(bins=bins, min=-range_max, max=range_max, sigma=sigma)
        sample = torch.randn(target_num, 1)
        h = self.gausshist(sample[:, 0].to(device))
        self.target = _normalise_target(h)

    def loss(self, x):
        total, _, _ = ops.latent_losses(x, max(x.shape[0], 2), self.target, 0.0, 0.0, 1.0, self.bins, self.range_max,
                                        self.sigma)
        return total


def _normalise_target(h):
    # h / sum(h) + 1e-8 on a 50-element device vector (constructor-time, once)
    return (h / h.sum() + 1e-8).contiguous()


def class_encode(label, device, ref_class):
    """Rows of the reference label table (one-hot by default) for each sample   (util.py:205-234)."""
    table = torch.tensor(np.asarray(ref_class), dtype=torch.float32)
    if torch.is_tensor(label) and label.is_cuda:
        # labels already on the device: look the rows up there (a .cpu() here would stall the host on the whole stream)
        from .model import host_to_device
        return host_to_device(table, label.device)[label.detach().long()].view(-1, table.shape[1]).to(device)
    idx = label.detach().cpu().long() if torch.is_tensor(label) else torch.as_tensor(label).long()
    from .model import host_to_device
    return host_to_device(table[idx].view(-1, table.shape[1]), device)


def get_target(label, classes, to_tensor=False, to_cuda=False, whole=False, shuffle=True):
    """Per sample, every class different from its source label, row-shuffled with numpy's global RNG
    (util.py:268-319).  Host logic; the notebooks take column 0 as the target label."""
    if torch.is_tensor(label):
        label = label.to("cpu").detach().numpy()
    label = np.asarray(label)
    n_cls = len(classes)
    grid = np.tile(np.arange(n_cls), (label.shape[0], 1))
    if whole:
        target = grid
    else:
        keep = grid != label[:, None]
        target = grid[keep].reshape(-1, n_cls - 1)
    if shuffle:
        for i in range(target.shape[0]):
            np.random.shuffle(target[i, :])
    if to_tensor:
        target = torch.Tensor(target)
        if to_cuda:
            target = target.to("cuda")
    return target


def weights_init(m):
    """The reference's initialiser never fires: it looks for lower-case 'conv'/'linear'/'batchnorm' in
    class names such as 'Conv2d' (util.py:193-203), so networks keep PyTorch's default init.  Kept as the
    same no-op so ``net.apply(weights_init)`` in the notebooks behaves identically."""
    return None


def load_classifier(net, classifier_path, device):
    """Load pre-trained Encoder_classifier weights into an Encoder with strict=False (util.py:236-266)."""
    state = torch.load(classifier_path, map_location=device)
    print(net.load_state_dict(state, strict=False))
    return net

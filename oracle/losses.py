"""Loss functions of the SRGAN train step, restated on CPU tensors.

TEST INFRASTRUCTURE (see oracle/__init__.py).

* ``lsgan``            <- get_loss_D            pyfiles/util.py:457-462 (criterion = nn.MSELoss)
* ``class_mse``        <- get_domainloss_D      pyfiles/util.py:464-468
* ``corrcoef`` / ``corr_loss``  <- corrcoef / corrcoef_loss   pyfiles/util.py:470-517
* ``soft_histogram`` / ``HistogramImitation`` <- GaussianHistogram / histogram_imitation
                                                 pyfiles/util.py:521-553
* ``batch_kl``         <- pyfiles/util_notebook.py:644-650
* ``conventional_kl``  <- pyfiles/util_notebook.py:630-634
* ``one_hot_rows``     <- class_encode          pyfiles/util.py:205-234
"""
import math

import numpy as np
import torch


def lsgan(outputs, target):
    """mean over scales of mean((o - target)^2)."""
    total = 0.0
    for o in outputs:
        total = total + ((o - target) ** 2).mean()
    return total / len(outputs)


def class_mse(class_probs, onehot):
    total = 0.0
    for q in class_probs:
        total = total + ((q - onehot) ** 2).mean()
    return total / len(class_probs)


def corrcoef(x):
    """Row-wise Pearson matrix of x [d, n]; np.corrcoef semantics incl. the +-1 clamp."""
    xm = x - x.mean(dim=1, keepdim=True)
    cov = xm @ xm.t() / (x.shape[1] - 1)
    sd = torch.sqrt(torch.diag(cov))
    r = cov / sd[None, :] / sd[:, None]
    return torch.clamp(r, -1.0, 1.0)


def corr_loss(m):
    d = m.shape[0]
    r = corrcoef(m)
    return (r - torch.eye(d, dtype=m.dtype)).abs().sum() / (d * (d - 1))


def hist_centers(bins=50, lo=-10.0, hi=10.0, dtype=torch.float32):
    delta = float(hi - lo) / float(bins)
    return float(lo) + delta * (torch.arange(bins).to(dtype) + 0.5), delta


def soft_histogram(x, bins=50, lo=-10.0, hi=10.0, sigma=0.2):
    """h_k = sum_n exp(-0.5 ((x_n - c_k)/sigma)^2) / (sigma sqrt(2 pi)) * delta, x is 1-D."""
    centers, delta = hist_centers(bins, lo, hi, x.dtype)
    d = x[None, :] - centers[:, None]
    k = torch.exp(-0.5 * (d / sigma) ** 2) / (sigma * np.sqrt(np.pi * 2)) * delta
    return k.sum(dim=1)


def analytic_hist_target(bins=50, range_max=10.0, sigma=0.2, dtype=torch.float32):
    """RNG-free stand-in for the sampled target: mass of N(0, 1+sigma^2) at the bin centres
    (SURVEY.md Appendix E).  Portable to the GPU box, where the reference RNG draw is unavailable."""
    centers, delta = hist_centers(bins, -range_max, range_max, torch.float64)
    v = 1.0 + sigma * sigma
    t = torch.exp(-centers ** 2 / (2 * v)) / math.sqrt(2 * math.pi * v) * delta
    return (t / t.sum() + 1e-8).to(dtype)


class HistogramImitation:
    """KL(target || soft-hist(x[:, j])) summed over the latent dims j.

    ``target=None`` reproduces the reference constructor: it draws
    ``torch.randn(target_num, 1)`` from the global CPU generator (util.py:543).
    """

    def __init__(self, bins=50, range_max=10, sigma=0.2, target_num=100000, target=None):
        self.bins, self.range_max, self.sigma = bins, float(range_max), sigma
        if target is None:
            sample = torch.randn(target_num, 1)
            h = soft_histogram(sample[:, 0], bins, -self.range_max, self.range_max, sigma)
            target = h / h.sum() + 1e-8
        self.target = target

    def loss(self, x):
        total = 0.0
        t = self.target.to(x.dtype)
        for j in range(x.shape[1]):
            h = soft_histogram(x[:, j], self.bins, -self.range_max, self.range_max, self.sigma)
            p = h / h.sum() + 1e-8
            total = total + (t * (t.log() - p.log())).sum()
        return total


def batch_kl(mu, n_batch):
    """-0.5 sum(1 + log v - m^2 - v), v = Var_unbiased(mu, 0) * n/(n-1) (double correction kept)."""
    var = torch.var(mu, dim=0) * n_batch / (n_batch - 1)
    mean = torch.mean(mu, dim=0)
    return -0.5 * torch.sum(1 + torch.log(var) - mean ** 2 - var)


def conventional_kl(mu, logvar):
    return -0.5 * torch.sum(1 + logvar - mu ** 2 - logvar.exp())


def one_hot_rows(label, ref_label, dtype=torch.float32):
    table = torch.as_tensor(np.asarray(ref_label), dtype=dtype)
    return table[label.cpu().long()].view(-1, table.shape[1])

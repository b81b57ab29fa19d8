"""GPU: the evaluation path (srgan_amd.evaluation: VGG19-bn features on the conv kernels, PRDC kernels) against the CPU
oracle (oracle/evaluation.py; parity unpinned vs torchvision / prdc, which are absent -- see the oracle's header)."""
import numpy as np
import pytest
import torch

from oracle import evaluation as oe
from tests.common import close

pytestmark = pytest.mark.gpu


def test_maxpool2_vs_torch():
    from srgan_amd.evaluation import maxpool2
    for shape in ((2, 8, 6, 10), (3, 64, 14, 14), (1, 4, 7, 9)):
        x = torch.randn(*shape, generator=torch.Generator().manual_seed(sum(shape)))
        y = maxpool2(x.cuda())
        assert torch.equal(y.cpu().contiguous(), torch.nn.functional.max_pool2d(x, 2, 2))


@pytest.mark.parametrize("div,batch,tol", [(8, 3, 2e-4), (1, 2, 2e-4)])
def test_vgg19_bn_features_vs_oracle(div, batch, tol):
    """div = 8: every layer at 1/8 width (ragged channel tiles); div = 1: the real network, 19.6 GFLOP per image: the 3x3
    layers run on F(4x4,3x3) / F(2x2,3x3) Winograd (14x14 maps) / the 3-channel first layer on the implicit GEMM, with the
    folded BatchNorm bias and the ReLU in their epilogues."""
    from srgan_amd import evaluation as he
    P = oe.fill(oe.vgg19_bn_spec(1000, div), 7)
    m = he.VGG19_bn(1000, div)
    m.load_state_dict(P)
    m.cuda().eval()
    x = torch.randn(batch, 3, 224, 224, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        want_f, want_s = oe.vgg19_bn_features(P, x), oe.vgg19_bn_scores(P, x)
    got_f = he.vgg_model(m).get(x.cuda(), "feature")
    got_s = he.vgg_model(m).get(x.cuda(), "score")
    assert got_f.shape == want_f.shape == (batch, 4096 // div)
    close(got_f, want_f, tol, what="vgg feature")
    close(got_s, want_s, tol, what="vgg score")
    with pytest.raises(NotImplementedError):
        m.train()
        m.feature(x.cuda())


@pytest.mark.parametrize("n,m,d,k", [(300, 257, 4096, 5), (64, 64, 7, 1), (5, 130, 33, 3), (1000, 900, 128, 5)])
def test_prdc_kernels_vs_oracle(n, m, d, k):
    from srgan_amd import _lib, ops
    from srgan_amd.evaluation import compute_prdc
    rng = np.random.RandomState(n + m)
    real = rng.randn(n, d).astype(np.float32)
    fake = (rng.randn(m, d) * 1.1 + 0.15).astype(np.float32)
    # distances and radii
    lib = _lib.load()
    x, y = torch.from_numpy(real).cuda(), torch.from_numpy(fake).cuda()
    dist = torch.empty(n, m, device="cuda")
    _lib.check(lib.srgan_pairwise_dist(ops._ptr(x), n, ops._ptr(y), m, d, ops._ptr(dist), ops._stream()), "pairwise_dist")
    want = oe.pairwise_distance(real, fake)
    assert np.abs(dist.cpu().numpy() - want).max() <= 5e-7 * want.max()        # chunked + compensated fp32 sum vs float64
    self_d = torch.empty(n, n, device="cuda")
    _lib.check(lib.srgan_pairwise_dist(ops._ptr(x), n, ops._ptr(x), n, d, ops._ptr(self_d), ops._stream()), "pairwise_dist")
    assert float(self_d.diagonal().abs().max()) == 0.0                     # exact zeros: (x - x)^2
    rad = torch.empty(n, device="cuda")
    _lib.check(lib.srgan_kth_smallest_rows(ops._ptr(self_d), n, n, min(k + 1, n), ops._ptr(rad), ops._stream()), "kth")
    got_r = rad.cpu().numpy()
    srt = np.sort(self_d.cpu().numpy(), axis=1)[:, min(k + 1, n) - 1]
    assert np.array_equal(got_r, srt)                                       # exact selection on the same matrix
    if k + 1 <= n and k + 1 <= m:
        got, ref = compute_prdc(real, fake, k), oe.compute_prdc(real, fake, k)
        # integer set statistics; a fp32-vs-fp64 tie in a distance comparison may move one count
        assert abs(got["precision"] - ref["precision"]) <= 1.0 / m + 1e-7 and abs(got["recall"] - ref["recall"]) <= 1.0 / n + 1e-7
        assert abs(got["coverage"] - ref["coverage"]) <= 1.0 / n + 1e-7 and abs(got["density"] - ref["density"]) <= 2.0 / (k * m) + 1e-7
        same = compute_prdc(real, real, k)
        assert same["precision"] == 1.0 and same["recall"] == 1.0 and same["coverage"] == 1.0 and abs(same["density"] - 1.0) < 1e-6
        far = compute_prdc(real, real + 100.0, k)
        assert far == dict(precision=0.0, recall=0.0, density=0.0, coverage=0.0)


def test_kth_value_with_ties_and_errors():
    from srgan_amd import _lib, ops
    lib = _lib.load()
    row = torch.tensor([[5.0, 1.0, 3.0, 1.0, 1.0, 7.0] + [9.0] * 200], device="cuda")
    out = torch.empty(1, device="cuda")
    for k, want in ((1, 1.0), (3, 1.0), (4, 3.0), (5, 5.0), (6, 7.0), (7, 9.0), (16, 9.0)):
        _lib.check(lib.srgan_kth_smallest_rows(ops._ptr(row), 1, row.shape[1], k, ops._ptr(out), ops._stream()), "kth")
        assert float(out) == want, (k, float(out))
    assert lib.srgan_kth_smallest_rows(ops._ptr(row), 1, row.shape[1], 17, ops._ptr(out), ops._stream()) != 0
    assert b"must lie in" in lib.srgan_last_error()


def test_gan_evaluation_get_prdc_end_to_end():
    """GAN_evaluation("vgg-initialization").get_prdc(true, pred) (evaluation.py:98-110) against the same pipeline on the oracle:
    8-bit rendering + the two PIL resizes + ImageNet normalisation, VGG19-bn features at full width, PRDC with k = 5."""
    from srgan_amd import evaluation as he
    torch.manual_seed(11)
    ev = he.GAN_evaluation("vgg-initialization", "cuda")
    P = {k: v.detach().cpu().clone() for k, v in ev.model.model.state_dict().items()}
    g = torch.Generator().manual_seed(5)
    true = torch.rand(12, 3, 128, 128, generator=g) * 2 - 1
    pred = (torch.rand(9, 3, 128, 128, generator=g) * 2 - 1) * 0.8
    pt = ev.preprocess(true)
    assert pt.shape == (12, 3, 224, 224) and torch.equal(pt, oe.preprocess(true))
    got = ev.get_prdc(true, pred, nearest_k=5)
    with torch.no_grad():
        f1 = oe.vgg19_bn_features(P, oe.preprocess(true)).numpy()
        f2 = oe.vgg19_bn_features(P, oe.preprocess(pred)).numpy()
    close(ev.get_feature(pt), f1, 2e-4, what="features of the real set")
    ref = oe.compute_prdc(f1, f2, 5)
    for key in ref:
        assert abs(got[key] - ref[key]) <= 1.0 / 9 + 1e-6, (key, got, ref)
    init = he.evaluation_init(["vgg-initialization"], (0, 1), {"precision": None, "recall": None})
    assert init["vgg-initialization"][1][0] == {"precision": [], "recall": []}

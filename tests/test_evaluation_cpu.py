"""CPU: the evaluation oracle (oracle/evaluation.py).  torchvision and prdc are absent from the image and the
reference holds no fixture for this path (PARITY UNPINNED, see the oracle's header); what can be checked here is that the
restated arithmetic agrees with independent installed implementations (scikit-learn's pairwise distances, a brute-force PRDC)
and with the known answers of the published definition."""
import numpy as np
import torch

from oracle import evaluation as oe


def test_pairwise_distance_equals_sklearn():
    from sklearn.metrics import pairwise_distances
    rng = np.random.RandomState(0)
    x, y = rng.randn(40, 16).astype(np.float32), (rng.randn(50, 16) * 1.2 + 0.3).astype(np.float32)
    np.testing.assert_allclose(oe.pairwise_distance(x, y), pairwise_distances(x, y, metric="euclidean"), rtol=0, atol=2e-6)
    d = oe.pairwise_distance(x)
    assert np.all(np.diag(d) == 0) and np.allclose(d, d.T)


def _prdc_brute(real, fake, k):
    """The definition of Naeem et al. spelled out with Python loops."""
    def radius(s, i):
        return sorted(float(np.linalg.norm(s[i].astype(np.float64) - s[j].astype(np.float64))) for j in range(len(s)))[k]
    rr = [radius(real, i) for i in range(len(real))]
    rf = [radius(fake, j) for j in range(len(fake))]
    d = [[float(np.linalg.norm(real[i].astype(np.float64) - fake[j].astype(np.float64))) for j in range(len(fake))] for i in range(len(real))]
    n, m = len(real), len(fake)
    precision = np.mean([any(d[i][j] < rr[i] for i in range(n)) for j in range(m)])
    recall = np.mean([any(d[i][j] < rf[j] for j in range(m)) for i in range(n)])
    density = np.mean([sum(d[i][j] < rr[i] for i in range(n)) for j in range(m)]) / k
    coverage = np.mean([min(d[i]) < rr[i] for i in range(n)])
    return dict(precision=precision, recall=recall, density=density, coverage=coverage)


def test_compute_prdc_known_answers_and_brute_force():
    rng = np.random.RandomState(3)
    real, fake = rng.randn(30, 8).astype(np.float32), (rng.randn(37, 8) * 0.8 + 0.4).astype(np.float32)
    got, want = oe.compute_prdc(real, fake, 3), _prdc_brute(real, fake, 3)
    for key in want:
        assert abs(got[key] - want[key]) < 1e-12, (key, got[key], want[key])
    same = oe.compute_prdc(real, real, 5)
    assert same["precision"] == 1.0 and same["recall"] == 1.0 and same["coverage"] == 1.0 and abs(same["density"] - 1.0) < 1e-12
    far = oe.compute_prdc(real, real + 100.0, 5)       # disjoint supports
    assert far == dict(precision=0.0, recall=0.0, density=0.0, coverage=0.0)
    assert oe.kth_value(np.array([[5.0, 1.0, 3.0, 1.0]]), 2)[0] == 1.0 and oe.kth_value(np.array([[5.0, 1.0, 3.0, 1.0]]), 3)[0] == 3.0


def test_vgg19_bn_layout_matches_torchvision_keys_and_the_host_mirror():
    """118 state_dict entries in torchvision's order: convs at features.{0,3,7,10,14,...}, their BatchNorm one index later,
    classifier.{0,3,6}; 143.7 M parameters at full width (torchvision's published count for vgg19_bn: 143,678,248)."""
    from srgan_amd import evaluation as he
    spec = oe.vgg19_bn_spec(1000)
    assert len(spec) == 118 and list(spec)[:2] == ["features.0.weight", "features.0.bias"]
    assert [k for k in spec if k.endswith(".weight") and len(spec[k]) == 4][:4] == [f"features.{i}.weight" for i in (0, 3, 7, 10)]
    n_params = sum(int(np.prod(s)) for k, s in spec.items() if not k.endswith(("running_mean", "running_var", "num_batches_tracked")))
    assert n_params == 143678248
    small = he.VGG19_bn(num_classes=4, width_div=16)
    sspec = oe.vgg19_bn_spec(4, 16)
    assert list(small.state_dict().keys()) == list(sspec.keys())
    assert all(tuple(v.shape) == tuple(sspec[k]) for k, v in small.state_dict().items())
    torch.manual_seed(0)
    m = he.VGG19_bn(width_div=16)          # torchvision's initialisation: conv biases 0, BN (1, 0), Linear N(0, 0.01)
    assert float(m.features[0].bias.abs().max()) == 0 and float(m.features[1].weight.min()) == 1
    assert abs(float(m.classifier[0].weight.std()) - 0.01) < 1e-3


def test_oracle_vgg_equals_an_nn_module_stack():
    """The functional oracle against the same network assembled from torch.nn modules (eval mode)."""
    import torch.nn as nn
    spec = oe.vgg19_bn_spec(10, 16)
    P = oe.fill(spec, 2)
    layers, cin = [], 3
    for v in oe.VGG19_CFG:
        if v == "M":
            layers.append(nn.MaxPool2d(2, 2))
        else:
            layers += [nn.Conv2d(cin, v // 16, 3, padding=1), nn.BatchNorm2d(v // 16), nn.ReLU()]
            cin = v // 16
    feats = nn.Sequential(*layers)
    cls = nn.Sequential(nn.Linear(cin * 49, 256), nn.ReLU(), nn.Dropout(), nn.Linear(256, 256), nn.ReLU(), nn.Dropout(), nn.Linear(256, 10))
    feats.load_state_dict({k[len("features."):]: v for k, v in P.items() if k.startswith("features.")})
    cls.load_state_dict({k[len("classifier."):]: v for k, v in P.items() if k.startswith("classifier.")})
    feats.eval(), cls.eval()
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        h = torch.flatten(nn.AdaptiveAvgPool2d((7, 7))(feats(x)), 1)
        want_f, want_s = cls[:6](h), cls(h)
        got_f, got_s = oe.vgg19_bn_features(P, x), oe.vgg19_bn_scores(P, x)
    assert torch.allclose(got_f, want_f, atol=1e-5) and torch.allclose(got_s, want_s, atol=1e-5)
    assert float(got_f.abs().max()) > 1e-3

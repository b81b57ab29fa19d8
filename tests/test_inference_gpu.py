"""GPU: the inference path (get_samples, image_from_output, forward-only hipGraph) against outputs of the reference's
``get_samples`` run on tier-T networks (tests/golden/inference_T.npz)."""
import os

import numpy as np
import pytest
import torch

from tests.common import build_hip_nets, close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "inference_T.npz"))


def _dataset(gold):
    return [(torch.from_numpy(gold["images"][i]), int(gold["labels"][i])) for i in range(3)]


def test_get_samples_tensor_mode(gold):
    from srgan_amd.inference import get_samples
    G, _, E = build_hip_nets("T")
    data, label = get_samples(G, E, _dataset(gold), 1, latent=gold["latent"], classes=(0, 1, 2, 3), ref_label=np.eye(4), ndim=8,
                              image_type="tensor", batch=2, device="cuda")
    assert np.array_equal(label["source"], gold["source_label"])
    close(data["source"], gold["source"], 1e-6, what="source")
    for c in range(4):
        assert len(label["latent"][c]) == 3                      # 5 codes in chunks of 2
        close(data["target"][c], gold[f"target.{c}"], 2e-4, what=f"target {c}")
        close(np.concatenate(label["latent"][c], 0), gold[f"mu.{c}"], 2e-4, what=f"mu {c}")


def test_get_samples_pil_mode(gold):
    from srgan_amd.inference import get_samples
    G, _, E = build_hip_nets("T")
    data, _ = get_samples(G, E, _dataset(gold), 1, latent=gold["latent"][:2], classes=(0, 1), ref_label=np.eye(4), ndim=8,
                          image_type="pil", batch=32, device="cuda")
    assert np.array_equal(np.asarray(data["source"]), gold["pil.source"])
    got = np.stack([np.asarray(im) for im in data["target"][1]]).astype(np.int32)
    assert np.abs(got - gold["pil.target.1"].astype(np.int32)).max() <= 1      # 8-bit quantisation of a 2e-4-close tensor


def test_graphed_forward_replays_the_generator(gold):
    from srgan_amd.inference import GraphedForward
    G, _, E = build_hip_nets("T")
    G.eval(); E.eval()
    x = torch.from_numpy(gold["images"][:2]).cuda()
    code = torch.cat([torch.eye(4)[[1, 2]], torch.from_numpy(gold["latent"][:2])], 1).cuda()

    def fwd(img, c):                 # mu only: the full Encoder.forward draws its noise on the CPU (not capturable)
        y = G(img, c)
        return y, E.fcmean(E.features(y))

    with torch.no_grad():
        y_ref, mu_ref = fwd(x, code)
    g = GraphedForward(fwd, x, code)
    y, mu = g(x, code)
    torch.cuda.synchronize()
    close(y, y_ref, 1e-6, what="graphed G")
    close(mu, mu_ref, 1e-6, what="graphed E mu")
    x2 = torch.from_numpy(gold["images"][1:3]).cuda()
    with torch.no_grad():
        y2_ref, _ = fwd(x2, code)
    y2, _ = g(x2, code)
    torch.cuda.synchronize()
    close(y2, y2_ref, 1e-6, what="graphed G, new input")


def test_get_output_and_plot_vs_reference(golden_dir):
    """The sample sheet of the train loop (util_notebook.py:738-846; 05-train cell 24 calls it every third of an epoch) against
    the figure the reference itself draws for the same tier-T networks, seed and dataset: same panels in the same slots with the
    same titles, 8-bit images within +-1."""
    import matplotlib
    matplotlib.use("Agg")
    import torch.nn as nn
    from oracle import trainer as otrainer
    from srgan_amd.inference import get_output_and_plot
    from srgan_amd.trainer import SRGAN_training
    gold = np.load(os.path.join(golden_dir, "plot_T.npz"))
    G, D, E = build_hip_nets("T")
    sg = SRGAN_training([G, D, E], [None, None, None], [nn.MSELoss(), nn.MSELoss()], dict(otrainer.DEFAULT_LBD), 1, "cuda",
                        np.eye(4), 4, "mu", 8)
    torch.manual_seed(3)
    dataset = [(torch.rand(3, 128, 128) * 2 - 1, int(i % 4)) for i in range(3)]
    np.testing.assert_allclose([float(d[0].double().sum()) for d in dataset], gold["image_checksum"], rtol=1e-12)
    names = ["male, smiling", "male, not smiling", "female, smiling", "female, not smiling"]
    torch.manual_seed(5)
    fig = get_output_and_plot(sg, dataset, 1, [(0, 1, 2, 3), names], 3, "cuda")
    assert list(fig.get_size_inches()) == list(gold["figsize"])
    assert [ax.get_title() for ax in fig.axes] == [str(t) for t in gold["titles"]]
    slots = [list(ax.get_subplotspec().get_geometry()[:3]) for ax in fig.axes]
    assert slots == gold["slots"].tolist()
    got = np.stack([np.asarray(ax.images[0].get_array()) for ax in fig.axes]).astype(np.int32)[:, ::4, ::4]
    assert np.abs(got - gold["panels"].astype(np.int32)).max() <= 1

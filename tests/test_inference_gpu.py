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

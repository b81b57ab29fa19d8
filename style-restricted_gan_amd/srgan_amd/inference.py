"""Inference path of the reference: ``get_samples`` (pyfiles/util_notebook.py:858-949) and the tensor ->
PIL helper ``image_from_output`` (pyfiles/util.py:157-188) over the HIP modules, plus a forward-only hipGraph wrapper.

Same names, arguments and return structure as the reference, so the evaluation / plotting cells of the notebooks run
unchanged on networks built from ``srgan_amd.model``."""
import numpy as np
import torch

from .losses import class_encode

__all__ = ["image_from_output", "dic_init", "get_samples", "GraphedForward"]


def _numpy(x):
    return x.detach().to("cpu").numpy() if torch.is_tensor(x) else np.asarray(x)


def image_from_output(output):
    """[n, c, h, w] (or [c, h, w]) tensor -> list of PIL images: per-image min-max to [0, 1], * 256, clip, uint8."""
    from PIL import Image
    if len(output.shape) == 3:
        output = output.unsqueeze(0)
    arr = _numpy(output)
    images = []
    for a in arr:
        a = np.tile(np.transpose(a, axes=(1, 2, 0)), (1, 1, int(3 / a.shape[0])))
        lo, hi = a.min(axis=None, keepdims=True), a.max(axis=None, keepdims=True)
        a = (a - lo) / (hi - lo + 1e-8) * 2 ** 8
        a[a > 255] = 255
        images.append(Image.fromarray(np.uint8(a)))
    return images


def dic_init(get_edge=False):
    return {"source": [], "target": [], "recon": []}, {"source": [], "target": []}


@torch.no_grad()
def get_samples(netG, netE, dataset, index, latent=None, classes=tuple(range(4)), ref_label=None, ndim=8, scale=1,
                image_type="pil", batch=32, device="cuda", conventional_E=False):
    """One source image -> for every target class the generator outputs for each latent code (in chunks of ``batch``) and
    the encoder's mu of those outputs.  ``latent``: [num, ndim] array shared by all classes, or a list with one array per
    class.  Returns ``(data, label)`` exactly as the reference: ``data["source"]``, ``data["target"][cls]`` (list of PIL
    images or one tensor), ``label["source"]``, ``label["latent"][cls]`` (list of mu arrays, one per chunk)."""
    source = dataset[index][0]
    fixed_source_image = source.view(1, 3, source.shape[-2], source.shape[-1]).to(device)
    fixed_source_label = torch.tensor([dataset[index][1]])
    data, label = dic_init(False)
    label["source"] = _numpy(fixed_source_label)
    data["source"] = image_from_output(fixed_source_image)[0] if image_type == "pil" else fixed_source_image.to("cpu")[0]
    netG.eval()
    netE.eval()
    if isinstance(latent, list):
        latent_list = [torch.tensor(v, dtype=torch.float32).to(device) for v in latent]
    else:
        latent_list = [torch.tensor(latent, dtype=torch.float32).to(device)] * len(classes)
    num = latent_list[0].shape[0]
    label["latent"], data["target"] = {}, {}
    for cls in classes:
        label["latent"][cls], data["target"][cls] = [], []
        class_vector = class_encode(torch.tensor([cls]), device, ref_label)
        chunks = []
        for start in range(0, num, batch):
            z = latent_list[cls][start:start + batch, :]
            code = torch.cat([class_vector.repeat(z.shape[0], 1), z], 1)
            target_image = netG(fixed_source_image.repeat(z.shape[0], 1, 1, 1), code)
            if conventional_E:
                _, mu, _ = netE(target_image, class_vector.repeat(z.shape[0], 1))
            else:
                _, mu, _, _, _ = netE(target_image)
            label["latent"][cls].append(_numpy(mu))
            if image_type == "pil":
                data["target"][cls] += image_from_output(target_image)
            else:
                chunks.append(_numpy(target_image))
        if image_type == "tensor":
            data["target"][cls] = torch.Tensor(np.concatenate(chunks, axis=0))
    if image_type == "tensor":
        data["source"] = torch.Tensor(data["source"]).unsqueeze(0)
    return data, label


class GraphedForward:
    """Forward-only hipGraph of a module call with fixed input shapes: the launches of one ``fn(*inputs)`` are captured
    once and replayed from static buffers -- removes the host launch cost (~20 us x a few hundred kernels per generator
    pass) from repeated inference.  Weights are read from their live storage; re-capture after changing them only if the
    packed-weight cache was invalidated (an optimiser step)."""

    def __init__(self, fn, *example_inputs):
        self.static_in = [x.clone() for x in example_inputs]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(2):                       # warm-up: packs weights, sizes workspaces, outside the capture
                fn(*self.static_in)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.static_out = fn(*self.static_in)

    def __call__(self, *inputs):
        for dst, src in zip(self.static_in, inputs):
            dst.copy_(src)
        self.graph.replay()
        return self.static_out

"""Inference path of the reference: ``get_samples`` (pyfiles/util_notebook.py:858-949) and the tensor ->
PIL helper ``image_from_output`` (pyfiles/util.py:157-188) over the HIP modules, plus a forward-only hipGraph wrapper.

Same names, arguments and return structure as the reference, so the evaluation / plotting cells of the notebooks run
unchanged on networks built from ``srgan_amd.model``."""
import numpy as np
import torch

from .losses import class_encode

__all__ = ["image_from_output", "dic_init", "get_samples", "GraphedForward", "cuda2numpy", "cuda2cpu", "get_output_tensors",
           "get_output_and_plot"]


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


def cuda2numpy(x):
    """device (or host) tensor -> ndarray   (pyfiles/util.py:15-36)."""
    return x.detach().to("cpu").numpy()


def cuda2cpu(x):
    """device tensor -> host tensor   (pyfiles/util.py:38-59)."""
    return x.detach().to("cpu")


@torch.no_grad()
def get_output_tensors(sg, dataset, index, classes, random_sample_num=5, device="cuda"):
    """The seven ``G_transformation`` calls of ``get_output_and_plot`` (pyfiles/util_notebook.py:764-799) in the reference's
    order (the style noise comes from the CPU default generator, so the order fixes the result).  Returns host tensors:
    source, target / recon / identity under the source's own style, one translation per other class, and
    ``random_sample_num`` random-style targets, reconstructions and identities."""
    from .losses import get_target
    data = dataset[index]
    src = data[0].view(1, 3, data[0].shape[-2], data[0].shape[-1]).to(device)
    src_label = torch.tensor(data[1]).view(1,)
    tgt_labels = torch.tensor(get_target(src_label, classes, whole=False, shuffle=False))
    tgt = tgt_labels[:, 0:1]
    n = random_sample_num
    out = {"source": cuda2cpu(src), "target_labels": tgt_labels}
    out["target"] = cuda2cpu(sg.G_transformation(tgt, src, True, src)[0])
    out["target_random"] = cuda2cpu(sg.G_transformation(tgt.repeat(1, n), src.repeat(n, 1, 1, 1), False)[0])
    first = out["target_random"][0:1].to(device)
    out["recon"] = cuda2cpu(sg.G_transformation(src_label, first, True, src)[0])
    out["identity"] = cuda2cpu(sg.G_transformation(src_label, src, True, src)[0])
    out["translations"] = cuda2cpu(sg.G_transformation(tgt_labels, src.repeat(len(classes) - 1, 1, 1, 1), False)[0])
    out["recon_random"] = cuda2cpu(sg.G_transformation(src_label.repeat(n), first.repeat(n, 1, 1, 1), False)[0])
    out["identity_random"] = cuda2cpu(sg.G_transformation(src_label.repeat(n), src.repeat(n, 1, 1, 1), False)[0])
    return out


def get_output_and_plot(sg, dataset, index, class_info, random_sample_num=5, device="cuda"):
    """Sample sheet of the train notebooks (pyfiles/util_notebook.py:738-846; called every third of an epoch by 05-train
    cell 24): same arguments, same figure layout (4 columns, ``random_sample_num + 1`` rows, same titles), returns the
    matplotlib figure.  The images come from the HIP modules through ``sg.G_transformation``."""
    import matplotlib.pyplot as plt
    classes, label_discription = class_info
    t = get_output_tensors(sg, dataset, index, classes, random_sample_num, device)
    n = random_sample_num
    length, width = n + 1, 4
    fig = plt.figure(figsize=(5 * width, 5 * length))

    def panel(slot, tensor, title):
        ax = fig.add_subplot(length, width, slot)
        ax.imshow(image_from_output(tensor)[0])
        ax.set_title(title)

    panel(1, t["source"], "source")
    panel(2, t["target"], "target by source condition")
    panel(3, t["recon"], "recon by source condition")
    panel(4, t["identity"], "identity image by source condition")
    for i in range(len(classes) - 1):
        panel(4 * (i + 1) + 1, t["translations"][i:i + 1], label_discription[t["target_labels"][0][i]])
    for col, key, title in ((2, "target_random", "target by random latent"), (3, "recon_random", "recon by random latent"),
                            (4, "identity_random", "idt by random latent")):
        for i in range(n):
            panel(4 * (i + 1) + col, t[key][i:i + 1], title)
    return fig

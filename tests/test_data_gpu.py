"""GPU: the HIP input pipeline (crop + Pillow-exact resize + flip + ToTensor + MinMax) against the Pillow / numpy
restatement of the reference transform -- bit-exact (integer resample, IEEE fp32 afterwards)."""
import numpy as np
import pytest
import torch

from oracle import preprocess as opre

pytestmark = pytest.mark.gpu
pytest.importorskip("PIL.Image")


def _images(b, h, w, seed):
    rng = np.random.default_rng(seed)
    x = rng.integers(0, 256, (b, h, w, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    x[0] = np.stack([(yy * 255 // max(h - 1, 1)), (xx * 255 // max(w - 1, 1)), ((yy + xx) % 256)], -1).astype(np.uint8)   # smooth image
    if b > 1:
        x[1, :, :, :] = 77                                                 # constant image: max == min, the 1e-8 guard
    return x


@pytest.mark.parametrize("h,w,crop,size", [(218, 178, (178, 178), (128, 128)), (200, 190, (178, 178), (128, 128)),
                                           (64, 80, (48, 64), (32, 40)), (128, 128, (128, 128), (128, 128))])
def test_gpu_transform_is_bit_exact(h, w, crop, size):
    from srgan_amd.data import GpuTransform
    x = _images(5, h, w, seed=h * 7 + w)
    flips = [1, 0, 1, 1, 0]
    t = GpuTransform(crop=crop, size=size, p=0.5, minmax=True, mean0=True)
    out = t(torch.from_numpy(x), flips=flips)
    assert out.shape == (5, 3, size[0], size[1])
    got = out.cpu().numpy()
    for i in range(5):
        ref = opre.transform_pil(x[i], flips[i], crop, size, True, True)
        assert np.array_equal(got[i], ref), f"image {i}: max |diff| {np.abs(got[i] - ref).max()}"


def test_flip_flags_follow_random_horizontal_flip_rng():
    from srgan_amd.data import GpuTransform
    t = GpuTransform()
    torch.manual_seed(3)
    mine = t.draw_flips(16).tolist()
    torch.manual_seed(3)
    ref = [int(bool(torch.rand(1) < 0.5)) for _ in range(16)]
    assert mine == ref


def test_prefetch_loader_yields_transformed_batches_in_order():
    from srgan_amd.data import GpuTransform, PrefetchLoader
    x = _images(6, 218, 178, seed=1)
    batches = [(torch.from_numpy(x[0:2]), torch.tensor([0, 1])), (torch.from_numpy(x[2:4]), torch.tensor([2, 3])),
               (torch.from_numpy(x[4:6]), torch.tensor([1, 0]))]
    t = GpuTransform(p=0.0)
    seen = []
    for img, lab in PrefetchLoader(batches, t):
        torch.cuda.synchronize()
        seen.append((img.cpu().numpy(), lab.tolist()))
    assert [s[1] for s in seen] == [[0, 1], [2, 3], [1, 0]]
    for bi, (img, _) in enumerate(seen):
        for j in range(2):
            assert np.array_equal(img[j], opre.transform_pil(x[2 * bi + j], 0))

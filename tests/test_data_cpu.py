"""CPU: the resize tables of the GPU input pipeline against Pillow itself.  The tables are what the HIP
kernels consume; a numpy restatement of Pillow's two-pass fixed-point resample driven by them must reproduce
``Image.resize(..., BILINEAR)`` byte for byte."""
import numpy as np
import pytest
import torch

from oracle import preprocess as opre
from srgan_amd import data as hdata

PIL = pytest.importorskip("PIL.Image")


@pytest.mark.parametrize("in_size,out_size", [(178, 128), (178, 64), (100, 128), (64, 64), (218, 37)])
def test_tables_reproduce_pillow_bilinear(in_size, out_size):
    rng = np.random.default_rng(in_size * 1000 + out_size)
    img = rng.integers(0, 256, (in_size, in_size + 5, 3), dtype=np.uint8)
    bh, ch, _ = hdata.pil_bilinear_tables(in_size + 5, out_size)
    bv, cv, _ = hdata.pil_bilinear_tables(in_size, out_size + 3)
    ours = opre.resize_restated(img, bh, ch, bv, cv)
    ref = np.asarray(PIL.fromarray(img, "RGB").resize((out_size, out_size + 3), PIL.BILINEAR))
    assert np.array_equal(ours, ref)


def test_center_crop_box_matches_torchvision_rule():
    assert hdata.center_crop_box(218, 178, 178, 178) == (20, 0)
    assert hdata.center_crop_box(219, 181, 178, 178) == (20, 2)       # round-half-even of 20.5 and 1.5, as Python's round
    with pytest.raises(ValueError):
        hdata.center_crop_box(100, 100, 178, 178)


def test_facedataset_selection_matches_reference(golden_dir, tmp_path):
    """File lists / labels of srgan_amd.data.FaceDataset against the reference class run on the same synthetic label set
    (tests/golden/make_golden.py: golden_facedataset)."""
    import json, os, pickle
    gold = json.load(open(os.path.join(golden_dir, "facedataset.json")))
    lab = str(tmp_path / "labels") + os.sep
    os.makedirs(lab)
    for i, rows in enumerate(gold["label_files"]):
        with open(os.path.join(lab, "part%d.pkl" % i), "wb") as fh:
            pickle.dump(np.array(rows), fh)
    assert [list(t) for t in hdata.get_class_label(3)] == gold["class_label_3"]
    for case in gold["cases"]:
        for dt, want in case["splits"].items():
            ds = hdata.FaceDataset("IMG/", lab, None, case["dataset_label"], tuple(case["classes"]), dt, case["train_num"],
                                   case["val_num"], case["test_num"])
            assert ds.images == want["images"], (case["dataset_label"], dt)
            assert [int(v) for v in ds.labels] == want["labels"]
            assert len(ds) == len(want["images"])


def test_facedataset_decodes_uint8_for_the_gpu_transform(tmp_path):
    import os, pickle
    rng = np.random.default_rng(0)
    root = str(tmp_path / "img") + os.sep
    lab = str(tmp_path / "lab") + os.sep
    os.makedirs(root); os.makedirs(lab)
    rows = []
    for i in range(4):
        PIL.fromarray(rng.integers(0, 256, (218, 178, 3), dtype=np.uint8), "RGB").save(root + "%06d.png" % i)
        rows.append(["%06d.jpg" % i, "1" if i % 2 else "-1"])
    with open(lab + "a.pkl", "wb") as fh:
        pickle.dump(np.array(rows), fh)
    ds = hdata.FaceDataset(root, lab, None, {"class": [1], "delete": [], "existed": []}, (0, 1), "train", 10, 0, 0)
    img, label = ds[0]
    assert img.dtype == torch.uint8 and tuple(img.shape) == (218, 178, 3) and label == 0
    assert np.array_equal(img.numpy(), np.asarray(PIL.open(ds.images[0]).convert("RGB")))

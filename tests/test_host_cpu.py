"""CPU: host-side logic of srgan_amd that needs no GPU -- module tree / state_dict layout / default
init parity with the reference, error behaviour, host helpers, optimiser plumbing."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import params
from srgan_amd import dp, losses, model
from srgan_amd._lib import SrganHipError


def shapes_of(net):
    return [[k, list(v.shape)] for k, v in net.state_dict().items()]


def test_state_dict_layout_matches_reference(golden_dir):
    ref = json.load(open(os.path.join(golden_dir, "shapes.json")))
    G = model.SingleGenerator(3, 64, 2, 2, 6, "instance", num_con=12)
    D = model.SingleDiscriminator_solo_multi(3, 64, 2, 4, "instance", 4)
    E = model.Encoder(3, 8, 64, 4, "instance", 4, "cpu")
    assert shapes_of(G) == ref["F"]["G"] and len(ref["F"]["G"]) == 78
    assert shapes_of(D) == ref["F"]["D"] and len(ref["F"]["D"]) == 16
    assert shapes_of(E) == ref["F"]["E"] and len(ref["F"]["E"]) == 24
    assert shapes_of(model.SingleDiscriminator_original_multi(3, 64, 2, 4, "instance")) == ref["D_original_64"]
    assert shapes_of(model.Encoder_original(3, 8, 64, 4, "instance", 2, "cpu")) == ref["E_original_64"]
    assert shapes_of(model.SingleGenerator(3, 64, 2, 2, 6, "instance", num_con=10)) == ref["G_cfg1"]
    assert all(v.dtype == torch.float32 for v in G.state_dict().values())
    # Encoder_classifier keys = Encoder keys minus fcmean/fcvar (used by freeze_melt in 05-train cell 22)
    C = model.Encoder_classifier(3, 8, 64, 4, "instance", 4)
    assert list(C.state_dict().keys()) == [k for k in E.state_dict().keys() if not k.startswith(("fcmean", "fcvar"))]


def test_default_init_equals_reference_under_seed(golden_dir):
    ref = json.load(open(os.path.join(golden_dir, "shapes.json")))["init_seed7"]
    torch.manual_seed(7)
    nets = dict(G=model.SingleGenerator(3, 8, 2, 2, 2, "instance", num_con=12),
                D=model.SingleDiscriminator_solo_multi(3, 8, 2, 4, "instance", 4),
                E=model.Encoder(3, 8, 8, 4, "instance", 4, "cpu"))
    for name, net in nets.items():
        net.apply(losses.weights_init)          # no-op, as in the reference
        got = [[k, float(v.double().sum()), float(v.double().abs().sum())] for k, v in net.state_dict().items()]
        for (k1, s1, a1), (k2, s2, a2) in zip(got, ref[name]):
            assert k1 == k2
            assert abs(s1 - s2) < 1e-9 and abs(a1 - a2) < 1e-9, (name, k1)


def test_freeze_melt_pretrained_encoder_recipe():
    E = model.Encoder(3, 8, 8, 4, "instance", 4, "cpu")
    C = model.Encoder_classifier(3, 8, 8, 4, "instance", 4)
    keys = list(C.state_dict().keys())
    E.freeze_melt(keys, "freeze")
    trainable = [k for k, p in E.named_parameters() if p.requires_grad]
    assert trainable == ["fcmean.weight", "fcmean.bias", "fcvar.weight", "fcvar.bias"]
    assert sum(p.numel() for p in E.parameters() if p.requires_grad) == 2 * (8 * 128 + 8)
    E.freeze_melt(keys, "melt")
    assert all(p.requires_grad for p in E.parameters())
    missing = E.load_state_dict(C.state_dict(), strict=False)
    assert sorted(missing.missing_keys) == ["fcmean.bias", "fcmean.weight", "fcvar.bias", "fcvar.weight"]


def test_error_behaviour():
    with pytest.raises(NotImplementedError, match=r"normalization layer \[foo\] is not found"):
        model.get_norm_layer("foo")
    cb = model.CBINorm2d(8, num_con=4, affine=True)
    with pytest.raises(ValueError, match=r"expected 4D input \(got 3D input\)"):
        cb(torch.zeros(2, 8, 5), torch.zeros(2, 4))
    with pytest.raises(SrganHipError, match="no CPU fallback"):
        model.SingleGenerator(3, 4, 2, 2, 1, "instance", num_con=12)(torch.zeros(1, 3, 16, 16), torch.zeros(1, 12))
    with pytest.raises(NotImplementedError, match="MSELoss"):
        losses.get_loss_D([torch.zeros(1)], 1.0, nn.L1Loss())


def test_class_encode_and_get_target():
    lab = torch.tensor([2, 0, 3, 1])
    oh = losses.class_encode(lab, "cpu", np.eye(4))
    assert oh.dtype == torch.float32 and oh.tolist() == np.eye(4)[[2, 0, 3, 1]].tolist()
    ref = np.array([[1.0, 0.5], [0.0, 2.0], [3.0, 3.0]])
    assert losses.class_encode(torch.tensor([2, 1]), "cpu", ref).tolist() == [[3.0, 3.0], [0.0, 2.0]]
    np.random.seed(0)
    t = losses.get_target(lab, (0, 1, 2, 3))
    assert t.shape == (4, 3)
    for row, src in zip(t, lab.tolist()):
        assert sorted(row.tolist()) == sorted(set(range(4)) - {src})
    assert losses.get_target(lab, (0, 1, 2, 3), whole=True, shuffle=False).tolist() == [[0, 1, 2, 3]] * 4
    assert losses.get_target(lab, (0, 1, 2, 3), shuffle=False)[0].tolist() == [0, 1, 3]


def test_minmax_transform():
    x = torch.tensor([[0.0, 5.0], [10.0, 2.5]])
    y = model.MinMax(True)(x)
    assert float(y.min()) == pytest.approx(-1.0) and float(y.max()) == pytest.approx(1.0, abs=1e-6)
    assert float(model.MinMax(False)(x).min()) == 0.0


def test_dataparallel_wrapper_exposes_module():
    net = model.SingleDiscriminator_solo_multi(3, 4, 2, 4, "instance", 4)
    w = dp.DataParallel(net, [0, 1, 2, 3])
    assert w.module is net and dp.unwrap(w) is net and dp.unwrap(net) is net
    assert w.n_class == 4                                  # attribute pass-through
    assert list(w.module.state_dict().keys()) == list(net.state_dict().keys())


def test_trainer_signature_and_optimiser_setup():
    import inspect
    from srgan_amd.trainer import SRGAN_training
    sig = inspect.signature(SRGAN_training.__init__)
    assert list(sig.parameters)[1:] == ["net", "opt", "criterion", "lbd", "unrolled_k", "device", "ref_label",
                                        "batch_size", "encoded_feature", "ndim"]
    assert sig.parameters["batch_size"].default == 64 and sig.parameters["encoded_feature"].default == "latent"
    for m in ("opt_sche_initialization", "G_transformation", "update_D", "update_GandE", "UnrolledUpdate", "train"):
        assert callable(getattr(SRGAN_training, m))
    G = model.SingleGenerator(3, 4, 2, 2, 1, "instance", num_con=12)
    D = model.SingleDiscriminator_solo_multi(3, 4, 2, 4, "instance", 4)
    E = model.Encoder(3, 8, 4, 4, "instance", 4, "cpu")
    lbd = {"class": 1, "cycle": 5, "idt": 5, "reg": .5, "idt_reg": .5, "KL": 0, "batch_KL": 10, "corr_enc": 100, "hist": 0}
    sg = SRGAN_training([G, D, E], [None, None, None], [nn.MSELoss(), nn.MSELoss()], lbd, 5, "cpu", np.eye(4), 32, "mu", 8)
    sg.opt_sche_initialization()
    for opt in (sg.optG, sg.optD, sg.optE):
        assert opt.param_groups[0]["lr"] == 1e-4 and opt.param_groups[0]["betas"] == (0.5, 0.999)
    sg.scheG.step()
    assert sg.optG.param_groups[0]["lr"] == pytest.approx(0.95e-4)


def test_singlegan_trainer_signature():
    import inspect
    from srgan_amd.trainer import SingleGAN_training
    sig = inspect.signature(SingleGAN_training.__init__)
    assert list(sig.parameters)[1:] == ["net", "opt", "criterion", "lbd", "unrolled_k", "device", "ref_label", "ndim",
                                        "classes", "batch_size", "encoded_feature", "singleD"]
    assert sig.parameters["singleD"].default is False and sig.parameters["encoded_feature"].default == "latent"
    G = model.SingleGenerator(3, 4, 2, 2, 1, "instance", num_con=10)
    D = [model.SingleDiscriminator_original_multi(3, 4, 2, 4, "instance") for _ in range(2)]
    E = model.Encoder_original(3, 8, 4, 4, "instance", 2, "cpu")
    lbd = {"class": 0, "cycle": 5, "idt": 5, "reg": .5, "idt_reg": 0, "KL": .1, "batch_KL": 0, "corr_enc": 0, "hist": 0}
    sg = SingleGAN_training([G, D, E], [None, "ignored", None], [nn.MSELoss(), nn.MSELoss()], lbd, 1, "cpu", np.eye(2), 8, (0, 1))
    sg.opt_sche_initialization()
    assert isinstance(sg.optD, list) and len(sg.optD) == 2 and len(sg.scheD) == 2     # always rebuilt per domain


def test_image_from_output_matches_reference(golden_dir):
    """tensor -> PIL conversion of the inference helpers (util.py:157-188) against the reference's output."""
    import os
    from srgan_amd.inference import image_from_output
    gold = np.load(os.path.join(golden_dir, "inference_T.npz"))
    img = image_from_output(torch.from_numpy(gold["source"]))[0]
    assert np.array_equal(np.asarray(img), gold["pil.source"])


def test_get_target_draw_order_vs_reference(golden_dir):
    """``get_target`` (util.py:268-319) under np.random.seed against the reference's own output: same targets, and the global
    numpy generator left at the same position (one np.random.shuffle per row, in row order)."""
    import json
    from srgan_amd import losses as hl
    fx = json.load(open(os.path.join(golden_dir, "host_logic.json")))
    for c in fx["get_target"]:
        np.random.seed(c["seed"])
        t = hl.get_target(torch.tensor(c["labels"]), tuple(range(c["n_cls"])), whole=c["whole"], shuffle=c["shuffle"])
        assert np.asarray(t).tolist() == c["target"], c
        assert float(np.random.rand()) == c["next_rand"], c
        if not c["whole"]:
            assert all(c["labels"][i] not in row for i, row in enumerate(np.asarray(t).tolist()))
    c = fx["get_target_tensor"]
    np.random.seed(c["seed"])
    t = hl.get_target(torch.tensor(c["labels"]), (0, 1, 2), to_tensor=True)
    assert str(t.dtype) == c["dtype"] and t.tolist() == c["target"]


def test_load_classifier_pth_round_trip_vs_reference(golden_dir, tmp_path, capsys):
    """``load_classifier`` (util.py:236-266): an Encoder_classifier checkpoint written to a .pth and loaded into an Encoder
    with strict=False.  Same printed key report as the reference, same tensors afterwards (classifier keys replaced, fcmean /
    fcvar untouched), and the file this package writes has the reference's key names."""
    import json
    from oracle import params
    from srgan_amd import losses as hl, model
    fx = json.load(open(os.path.join(golden_dir, "host_logic.json")))["load_classifier"]
    spec_e = params.encoder_spec(3, 8, 4, 4, 4)
    spec_c = {k: v for k, v in spec_e.items() if not k.startswith(("fcmean", "fcvar"))}
    clf = model.Encoder_classifier(3, 8, 4, 4, "instance", 4)
    clf.load_state_dict(params.fill(spec_c, 40))
    assert list(clf.state_dict().keys()) == fx["clf_keys"]
    enc = model.Encoder(3, 8, 4, 4, "instance", 4, "cpu")
    enc.load_state_dict(params.fill(spec_e, 41))
    before = {k: v.clone() for k, v in enc.state_dict().items()}
    path = str(tmp_path / "classifier.pth")
    torch.save(clf.state_dict(), path)
    ret = hl.load_classifier(enc, path, "cpu")
    assert ret is enc
    assert capsys.readouterr().out.strip() == fx["printed"]
    sd = enc.state_dict()
    for k, (s, n) in fx["checksums"].items():
        assert abs(float(sd[k].double().sum()) - s) <= 1e-9 * max(abs(s), 1.0), k
        assert abs(float(sd[k].double().norm()) - n) <= 1e-9 * max(n, 1.0), k
    for k in fx["missing"]:
        assert torch.equal(sd[k], before[k])
    for k in fx["clf_keys"]:
        assert torch.equal(sd[k], clf.state_dict()[k])
    # freeze_melt on the loaded encoder: only fcmean / fcvar stay trainable (05-train cell 22)
    enc.freeze_melt(fx["clf_keys"], "freeze")
    assert sorted(k for k, p in enc.named_parameters() if p.requires_grad) == sorted(fx["missing"])


def test_notebook_cell_1_imports_resolve_through_the_compat_shims():
    """05-train cell 1 (notebook lines 29-34): ``sys.path.append("../pyfiles/")`` then four ``from <module> import ...``
    lines.  With the path entry pointed at srgan_amd/compat the same lines import the MI355X implementations."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "style-restricted_gan_amd")
    code = "\n".join([
        "import sys", f"sys.path.insert(0, {pkg!r})", f"sys.path.append({os.path.join(pkg, 'srgan_amd', 'compat')!r})",
        "from util import image_from_output, get_target, cuda2numpy, cuda2cpu, weights_init, load_classifier",
        "from dataset import get_class_label, FaceDataset",
        "from model import MinMax, SingleGenerator, SingleDiscriminator_solo_multi, Encoder, Encoder_classifier",
        "from util_notebook import SRGAN_training, get_output_and_plot",
        "import srgan_amd.trainer, srgan_amd.model",
        "assert SRGAN_training is srgan_amd.trainer.SRGAN_training and Encoder is srgan_amd.model.Encoder",
        "import torch", "t = torch.arange(4.0)", "assert cuda2numpy(t).tolist() == [0, 1, 2, 3] and cuda2cpu(t).device.type == 'cpu'"])
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1500:]


def test_non_mse_class_criterion_leaves_the_fused_paths():
    """The fused discriminator loss kernel is softmax + MSE.  Any other ``criterion_class`` must not be silently replaced by it:
    the trainer then takes the generic path, whose loss helpers refuse criteria without a HIP kernel (05-train cell 13 passes
    nn.MSELoss for both)."""
    import torch.nn as nn
    from srgan_amd import losses as hl, model
    from srgan_amd.trainer import SRGAN_training
    G = model.SingleGenerator(3, 4, 2, 2, 1, "instance", num_con=12)
    D = model.SingleDiscriminator_solo_multi(3, 4, 2, 4, "instance", 4)
    E = model.Encoder(3, 8, 4, 4, "instance", 4, "cpu")
    lbd = {"class": 1.0, "cycle": 5.0, "idt": 5.0, "reg": 0.5, "idt_reg": 0.5, "KL": 0.0, "batch_KL": 0.0, "corr_enc": 0.0, "hist": 0.0}
    ok = SRGAN_training([G, D, E], [None, None, None], [nn.MSELoss(), nn.MSELoss()], lbd, 1, "cpu", np.eye(4), 4, "mu", 8)
    assert ok._class_is_mse() and ok._fused_paths()
    for crit in (nn.L1Loss(), nn.MSELoss(reduction="sum"), nn.CrossEntropyLoss()):
        sg = SRGAN_training([G, D, E], [None, None, None], [nn.MSELoss(), crit], lbd, 1, "cpu", np.eye(4), 4, "mu", 8)
        assert not sg._class_is_mse() and not sg._fused_paths()
        with pytest.raises(NotImplementedError, match="nn.MSELoss"):
            hl.get_domainloss_D([torch.zeros(2, 4)], torch.zeros(2, 4), crit)
    assert ok._noise_kinds() == ["randn"] + ["normal"] * 2 + ["normal"] * 3

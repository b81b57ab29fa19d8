#!/usr/bin/env python3
"""Host-logic fixtures from the REFERENCE (run in the build container, needs /root/reference; writes numbers only):

* ``get_target`` (pyfiles/util.py:268-319) under ``np.random.seed``: the per-row ``np.random.shuffle`` draw order;
* ``load_classifier`` (pyfiles/util.py:236-266): a ``.pth`` written from the reference's ``Encoder_classifier`` loaded into
  the reference's ``Encoder`` with strict=False -- which keys are reported missing / unexpected, and per-tensor checksums
  of the encoder afterwards (weights come from the build-owned deterministic fill, so both sides can regenerate them).

Run from the repo root:  python tests/golden/make_host_golden.py
"""
import contextlib
import io
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
from oracle import params as oparams  # noqa: E402

_stub = types.ModuleType("prdc")
_stub.compute_prdc = lambda *a, **k: None
sys.modules["prdc"] = _stub
import matplotlib  # noqa: E402
matplotlib.use("Agg")
sys.path.insert(0, "/root/reference/pyfiles")
import model as ref_model            # noqa: E402
import util as ref_util              # noqa: E402

out = {}

# ---- get_target ------------------------------------------------------------------------------------
cases = []
for seed, labels, n_cls, whole, shuffle in ((0, [0, 1, 2, 3, 3, 2, 1, 0, 2, 2], 4, False, True),
                                            (7, [1, 0, 1, 1, 0], 2, False, True),
                                            (3, [4, 0, 2, 1, 3, 3], 5, False, True),
                                            (5, [0, 1, 2, 3], 4, True, True),
                                            (9, [2, 0, 1], 3, False, False)):
    np.random.seed(seed)
    t = ref_util.get_target(torch.tensor(labels), tuple(range(n_cls)), whole=whole, shuffle=shuffle)
    after = float(np.random.rand())           # the global generator's position after the call
    cases.append(dict(seed=seed, labels=labels, n_cls=n_cls, whole=whole, shuffle=shuffle, target=np.asarray(t).tolist(),
                      next_rand=after))
np.random.seed(11)
t = ref_util.get_target(torch.tensor([0, 2, 1]), (0, 1, 2), to_tensor=True)
out["get_target"] = cases
out["get_target_tensor"] = dict(seed=11, labels=[0, 2, 1], dtype=str(t.dtype), target=t.tolist())

# ---- load_classifier -------------------------------------------------------------------------------
spec_e = oparams.encoder_spec(3, 8, 4, 4, 4)
spec_c = {k: v for k, v in spec_e.items() if not k.startswith(("fcmean", "fcvar"))}
clf = ref_model.Encoder_classifier(3, 8, 4, 4, "instance", 4)
clf.load_state_dict(oparams.fill(spec_c, 40))
enc = ref_model.Encoder(3, 8, 4, 4, "instance", 4, "cpu")
enc.load_state_dict(oparams.fill(spec_e, 41))
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "classifier.pth")
    torch.save(clf.state_dict(), path)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        ret = ref_util.load_classifier(enc, path, "cpu")
    assert ret is enc
    res = enc.load_state_dict(torch.load(path, map_location="cpu"), strict=False)
out["load_classifier"] = dict(
    printed=buf.getvalue().strip(), missing=list(res.missing_keys), unexpected=list(res.unexpected_keys),
    clf_keys=list(clf.state_dict().keys()),
    checksums={k: [float(v.double().sum()), float(v.double().norm())] for k, v in enc.state_dict().items()})

with open(os.path.join(HERE, "host_logic.json"), "w") as f:
    json.dump(out, f, indent=1)
print("wrote host_logic.json:", out["load_classifier"]["printed"])

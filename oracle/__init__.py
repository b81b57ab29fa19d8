"""CPU oracle for the SRGAN G+D+E train-step path.

TEST INFRASTRUCTURE ONLY.  This package is a plain PyTorch-CPU (fp32/fp64)
restatement of the algorithm in the reference's ``pyfiles/model.py``,
``pyfiles/util.py:455-553`` and ``pyfiles/util_notebook.py:419-734``.  It is
used as the parity checker by ``tests/``, by ``__graft_entry__.smoke()`` and as
the ``cpu_baseline`` leg of ``bench.py`` -- and by nothing else.  The product
package (``style-restricted_gan_amd/srgan_amd``) never imports it and has no CPU
fallback: it raises if the HIP library is missing.  ``oracle/preprocess.py`` restates the input transform
through Pillow / numpy and is pinned by Pillow itself.

Parity status: PINNED.  The reference itself ships no tests / golden vectors
for this path, so the oracle is pinned against outputs of the
reference *run in the build container* (``tests/golden/make_golden.py`` imports
``/root/reference/pyfiles`` and writes small ``.npz`` fixtures); see
``tests/test_oracle_golden.py``.
"""
from . import params, nets, losses, trainer  # noqa: F401

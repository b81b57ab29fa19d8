"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/srgan_hip.h declares."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    import __graft_entry__
    __graft_entry__.build()
    from srgan_amd import _lib
    return _lib


def test_header_symbols_are_exported_and_bound(built_lib):
    header = open(os.path.join(ROOT, "include", "srgan_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(srgan_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 30
    lib = built_lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in srgan_hip.h but not exported"
    assert declared == set(built_lib.SIGNATURES), declared ^ set(built_lib.SIGNATURES)
    assert lib.srgan_abi_version() == 1


def test_invalid_arguments_are_reported_without_a_gpu(built_lib):
    import ctypes
    lib = built_lib.load()
    d = built_lib.ConvDesc(1, 8, 8, 4, 9, 9, 4, 3, 3, 1, 1, 0, 36, 9, 3, 1)   # wrong Ho/Wo
    assert lib.srgan_conv2d_workspace(ctypes.byref(d)) == 0
    assert b"do not match geometry" in lib.srgan_last_error()
    assert lib.srgan_adam_step(None, None, None, None, 10, 1e-3, 0.5, 0.999, 1e-8, 1, None) == -1
    with pytest.raises(built_lib.SrganHipError, match="adam"):
        built_lib.check(-1, "adam")


def test_product_refuses_cpu_tensors(built_lib):
    import torch
    from srgan_amd import ops
    with pytest.raises(built_lib.SrganHipError, match="no CPU fallback"):
        ops.conv2d(torch.zeros(1, 3, 8, 8), torch.zeros(4, 3, 3, 3))

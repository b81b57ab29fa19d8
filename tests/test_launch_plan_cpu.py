"""CPU: the launch descriptors the host code of csrc/ builds, and the code objects they launch, stay inside what an AQL dispatch
packet and a gfx950 compute unit allow -- for every layer of G / D / E at every BASELINE geometry, both compute modes.

VERDICT r5 item 2 (the `HSA_STATUS_ERROR_INVALID_PACKET_FORMAT` abort of `rocprofv3 --pmc` over the 256x256 fp32 step).  No GPU:
`tests/hip_shim/launch_shim.c` stands in for the HIP runtime under LD_PRELOAD and logs (kernel, grid, block, dynamic LDS) of
every launch `tests/hip_shim/drive_launches.py` provokes through the C ABI; the kernels' static LDS / scratch / kernarg sizes and
launch bounds come from the product library's code objects (`tests/isa_tools.py`)."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import isa_tools                                                               # noqa: E402

LDS_BYTES = 160 * 1024          # per workgroup, gfx950 (MI355X_MICROARCH.md)
KERNARG_MAX = 4096              # HIP's limit for by-value kernel arguments
SCRATCH_MAX = 4096              # bytes per lane we accept before calling it a runaway spill

# (name, image side, batch per GPU, discriminator depth): BASELINE.json configs[0..4]
GEOMETRIES = [("c0_64_b8", 64, 8, 4), ("c1_128_b32", 128, 32, 4), ("c3_128_b64", 128, 64, 4), ("c4_256_b16", 256, 16, 5)]


@pytest.fixture(scope="module")
def lib_path():
    import __graft_entry__
    __graft_entry__.build()
    from srgan_amd import _lib
    return _lib.LIB_PATH


@pytest.fixture(scope="module")
def descriptors(lib_path):
    return {k["name"]: k for k in isa_tools.kernel_descriptors(lib_path)}


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("shim") / "launch_shim.so")
    subprocess.run(["gcc", "-shared", "-fPIC", "-O1", "-o", so, os.path.join(HERE, "hip_shim", "launch_shim.c")], check=True)
    return so


def test_code_objects_fit_a_compute_unit(descriptors):
    """Static limits of every kernel in the product library, whatever launches it."""
    assert len(descriptors) >= 250
    for name, k in descriptors.items():
        assert k["lds"] <= LDS_BYTES, (name, k["lds"])
        assert k["kernarg"] <= KERNARG_MAX, (name, k["kernarg"])
        assert not k["dynamic_stack"], name
        assert k["scratch"] <= SCRATCH_MAX, (name, k["scratch"])
        assert 64 <= k["max_wg"] <= 1024, (name, k["max_wg"])
        assert k["vgpr"] + 0 <= 512 and k["agpr"] <= 256, (name, k["vgpr"], k["agpr"])
    # the matrix kernels that own the step do not grow their scratch unnoticed: a spill reload sits behind the in-order
    # memory counter (profiles/LOG.md, round 5).  Today: wino43_kernel<true> 116 bytes, <false> 52, wino42_kernel<2, true> 40.
    heavy = {n: k["scratch"] for n, k in descriptors.items() if k["scratch"] and "latent_losses" not in n}
    assert all(v <= 128 for v in heavy.values()), heavy
    assert len(heavy) <= 3, heavy


@pytest.mark.parametrize("name,H,B,ncls", GEOMETRIES)
def test_launch_descriptors_of_every_baseline_geometry(name, H, B, ncls, lib_path, descriptors, shim, tmp_path):
    log = str(tmp_path / "launches.log")
    env = dict(os.environ, LD_PRELOAD=shim, SRGAN_SHIM_LOG=log)
    env.pop("SRGAN_HIP_LIB", None)
    r = subprocess.run([sys.executable, os.path.join(HERE, "hip_shim", "drive_launches.py"), lib_path, str(H), str(B), str(ncls)],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    where, n, kernels = "", 0, set()
    for line in open(log):
        if line.startswith("#"):
            where = line[1:].strip()
            continue
        kname, gx, gy, gz, bx, by, bz, dyn = line.split()
        gx, gy, gz, bx, by, bz, dyn = map(int, (gx, gy, gz, bx, by, bz, dyn))
        ctx = (where, kname, (gx, gy, gz), (bx, by, bz))
        assert kname in descriptors, ctx
        k = descriptors[kname]
        n += 1
        kernels.add(kname)
        threads = bx * by * bz
        # AQL dispatch packet: 16-bit workgroup sizes, 32-bit grid sizes IN WORK-ITEMS, non-zero in every dimension
        assert min(gx, gy, gz, bx, by, bz) >= 1, ctx
        assert threads <= 1024 and threads <= k["max_wg"], ctx          # __launch_bounds__ of the code object
        assert threads % 64 == 0, ctx                                    # whole wavefronts (every kernel here assumes it)
        assert gx * bx < 2 ** 32 and gy * by < 2 ** 32 and gz * bz < 2 ** 32, ctx
        assert gx < 2 ** 31 and gy < 2 ** 16 and gz < 2 ** 16, ctx       # HIP's grid limits in workgroups
        assert gx * gy * gz * threads < 2 ** 40, ctx                     # nothing near a runaway grid
        assert k["lds"] + dyn <= LDS_BYTES, ctx
    assert n >= 500 and len(kernels) >= 40, (n, len(kernels))
    if H == 256:
        # the geometry the counter pass aborted on launches no kernel the 128x128 step does not also launch
        log128 = str(tmp_path / "l128.log")
        env["SRGAN_SHIM_LOG"] = log128
        subprocess.run([sys.executable, os.path.join(HERE, "hip_shim", "drive_launches.py"), lib_path, "128", "32", "4"], env=env,
                       check=True, capture_output=True, timeout=600)
        k128 = {ln.split()[0] for ln in open(log128) if not ln.startswith("#")}
        only256 = {re.sub(r"^_ZN5srgan", "", k)[:60] for k in kernels - k128}
        # two-pass norms of the 64x64 trunk maps and the fifth discriminator stage are allowed to differ; record what does
        assert len(only256) <= 12, sorted(only256)

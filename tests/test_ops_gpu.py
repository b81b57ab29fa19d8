"""GPU: every HIP op of libsrgan_hip.so (through the C ABI / ctypes) against a PyTorch-CPU fp32/fp64
restatement of the same op.  Tolerance: 2e-5 relative to the tensor's max magnitude for fp32 kernels
(exact-fp32 MFMA, different summation order), stated per test."""
import contextlib
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from srgan_amd import ops as _ops
    assert torch.cuda.is_available()
    return _ops


def close(a, b, rtol=2e-5, atol=1e-6):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = max(float(b.abs().max()), 1e-30)
    err = float((a - b).abs().max())
    assert err <= atol + rtol * scale, f"max err {err:.3e} vs scale {scale:.3e}"


def rnd(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g)


CONV_CASES = [
    # N, I, H, W, O, k, s, p, reflect, bias
    (2, 3, 20, 20, 8, 7, 1, 3, False, False),     # G first layer shape class (Cin=3)
    (2, 64, 16, 16, 128, 4, 2, 1, False, False),  # k4 s2 down conv (vector path)
    (2, 32, 12, 12, 64, 3, 1, 1, False, False),   # 3x3 residual conv class
    (1, 256, 8, 8, 256, 3, 1, 1, False, False),   # exact G res conv channels
    (2, 64, 14, 14, 3, 7, 1, 3, False, False),    # G last layer (Cout=3)
    (2, 3, 32, 32, 64, 4, 2, 1, False, False),    # D first layer
    (3, 128, 8, 8, 1, 4, 1, 1, False, True),      # D last_layer (Cout=1, bias)
    (3, 64, 8, 8, 4, 8, 1, 0, False, True),       # D classification head (valid 8x8)
    (2, 3, 33, 33, 16, 7, 2, 1, False, True),     # E first layer (k7 s2 p1, odd sizes)
    (2, 32, 9, 9, 64, 3, 1, 1, True, False),      # E reflect conv (vector path)
    (2, 4, 7, 7, 8, 3, 1, 1, True, False),        # reflect, generic-channel path
    (2, 8, 3, 3, 16, 3, 1, 1, True, False),       # reflect on a 3x3 map (both mirrors hit row 1)
    (2, 32, 6, 6, 64, 1, 1, 0, False, True),      # 1x1 shortcut with bias
    (1, 16, 10, 10, 32, 4, 2, 1, False, False),   # generic channels, stride 2
    (2, 160, 6, 6, 96, 3, 1, 1, False, False),    # non power-of-two channels (vector path, N mask)
    (2, 64, 64, 64, 3, 7, 1, 3, False, False),    # narrow-output direct kernels (G RGB head), exact tiles
    (2, 32, 67, 45, 1, 4, 1, 1, False, True),     # narrow-output, ragged tiles, Cout=1 + bias
    (3, 16, 40, 72, 4, 3, 1, 1, False, True),     # narrow-output, Cout=4, one channel chunk
    (4, 128, 48, 48, 256, 3, 1, 1, False, False), # 128x128 block tiles on both GEMM kernels, split-K > 1
    (8, 64, 64, 64, 64, 4, 2, 1, False, False),   # 128x64 tiles, many M tiles
    (4, 64, 32, 32, 128, 3, 1, 1, False, False),  # row-aligned weight-gradient path (Wo % 32 == 0), zero pad
    (2, 32, 32, 64, 64, 3, 1, 1, True, False),    # row-aligned weight-gradient path with reflect padding
    (2, 128, 64, 64, 128, 4, 2, 1, False, False), # row-aligned path, stride 2, 128x128 tiles, several splits
    (32, 256, 16, 16, 256, 3, 1, 1, False, False),# 8-wave 256x128 forward / dgrad tiles
    (3, 48, 7, 11, 40, 3, 1, 1, False, True),     # Winograd path: odd sizes, ragged tile / channel blocks, bias
    (2, 48, 5, 4, 72, 3, 1, 1, True, True),       # Winograd path: reflect padding, bias, 2 output-channel blocks
    (5, 64, 32, 32, 64, 3, 1, 1, False, False),   # Winograd path: tiles not a multiple of 64 per image boundary
    (8, 64, 18, 14, 128, 3, 1, 1, False, True),   # Winograd weight gradient: 63 tile positions (ragged last group), bias
    (4, 128, 17, 9, 128, 3, 1, 1, True, False),   # Winograd weight gradient: odd sizes (half tiles), reflect padding
    (16, 256, 16, 16, 128, 3, 1, 1, False, False),# Winograd weight gradient: several groups per split
    (8, 128, 40, 44, 128, 4, 2, 1, False, True),  # F(3x3,2x2) Winograd of a 4x4 stride-2 layer: fwd and input gradient, ragged 3x3 tiles
    (48, 128, 58, 62, 128, 4, 2, 1, False, True), # same + its Winograd weight gradient (needs >= 192 workgroups, >= 48 chunks), ragged tiles
    (32, 64, 32, 32, 64, 4, 2, 1, False, False),  # same, one cout tile, many images per tile-position group
    (64, 64, 64, 64, 128, 4, 2, 1, False, False), # D trunk at batch 64: Winograd weight gradient split over groups AND images (16 x 2)
    (50, 128, 32, 32, 256, 4, 2, 1, False, False),# same: 5 groups x 8 batch ranges, ragged last range (50 = 7 x 7 + 1)
    (2, 3, 24, 70, 64, 7, 1, 3, False, False),    # RGB 7x7 layer: input gradient through the narrow-output kernel (flipped filter)
    (2, 3, 12, 13, 32, 5, 1, 2, False, True),     # same route, generic narrow kernel (5x5), bias
    (2, 64, 20, 72, 3, 7, 1, 3, False, True),     # RGB head through the 7x1 row convolution + shift-add, bias, ragged rows
    (32, 64, 32, 32, 128, 3, 1, 1, False, True),  # F(4x4,3x3) Winograd by default dispatch (128 workgroups), bias
    (3, 64, 8, 12, 96, 3, 1, 1, False, True),     # F(4x4,3x3) when forced: 18 tiles (ragged block), 8 chunks, 3 channel blocks
    (2, 3, 40, 33, 128, 7, 1, 3, False, True),    # RGB-input 7x7 layer on the MFMA (LDS halo): ragged tiles, 2 channel blocks, bias
    (9, 3, 128, 128, 64, 7, 1, 3, False, False),  # RGB-input layer: MFMA weight gradient, 288 pixel tiles on 256 persistent workgroups
    (3, 64, 32, 64, 3, 7, 1, 3, False, False),    # RGB-output layer: MFMA weight gradient (swapped roles, flipped taps), 12 tiles
    (2, 3, 32, 40, 64, 7, 2, 1, False, True),     # E first layer at even sizes: input gradient as four stride-1 phase convs (3 / 4 taps)
    (3, 3, 33, 31, 32, 7, 2, 1, False, False),    # same, odd sizes (phase images of different heights / widths)
    (2, 4, 18, 22, 48, 6, 2, 2, False, False),    # same route: 4 input channels, 6x6 taps, pad 2
    (2, 3, 16, 16, 16, 5, 2, 0, False, False),    # same route: odd kernel, no padding
    (2, 256, 32, 32, 256, 3, 1, 1, False, False), # residual-trunk layer at full width (bf16 mode: LDS-resident patch kernel, C = 256)
    (1, 128, 8, 64, 128, 3, 1, 1, False, True),   # same kernel family: C = 128, two patches per row, bias
    (1, 256, 64, 64, 256, 3, 1, 1, False, False), # the 256x256 configuration's trunk map (64 x 64): 32 patches of one image
    (2, 64, 16, 64, 128, 4, 2, 1, False, False),  # 64 -> 128 down conv on an 8 x 32 output map (bf16 mode: input gradient on the transposed patch kernel)
    (1, 128, 8, 64, 256, 4, 2, 1, False, False),  # 128 -> 256 down conv, one output patch (same kernel, C = 256 reduce channels)
    (3, 64, 64, 128, 128, 4, 2, 1, False, True),  # same layer class on a wider map: 8 x 2 patches per image, bias
    (2, 64, 40, 70, 3, 7, 1, 3, False, True),     # round 3: RGB head on the 4x4x1 MFMA (direct, LDS halo): ragged 32 x 64 tiles, bias
    (3, 3, 128, 128, 64, 7, 1, 3, False, False),  # RGB input layer at full size: its input gradient on the same kernel (flipped, transposed filter)
    (2, 32, 64, 96, 3, 7, 1, 3, False, False),    # same kernel, 32 reduce channels (8 channel quads), 2 column tiles
    (4, 64, 62, 62, 128, 3, 1, 1, True, False),   # the encoder's own shapes (round 3): E.layers.0.cmp, 62 x 62, reflect padding
    (4, 128, 31, 31, 128, 3, 1, 1, True, True),   # E.layers.1 on 31 x 31, reflect, bias
    (8, 64, 30, 22, 64, 3, 1, 1, False, True),    # even but not multiple-of-4 map, zero padding, bias
    (6, 32, 15, 15, 64, 3, 1, 1, True, False),    # E.layers.2 map size, reflect
    (2, 96, 13, 18, 32, 3, 1, 1, True, True),     # odd sizes in both directions, 3 channel groups, one cout block, bias
    (16, 64, 66, 66, 64, 3, 1, 1, True, False),   # bf16 mode: 545 pixel tiles of igemm16_kernel on <= 512 persistent workgroups (reflect, 64 couts)
    (9, 256, 64, 16, 64, 4, 2, 1, False, False),  # round 5: 256 input channels keep the strided 4x4 / stride-2 form OFF F(4x4,2x2) (1024 reduce terms: 2.07e-5 there); its input gradient (64 reduce channels) takes it
    (4, 128, 32, 48, 192, 4, 2, 1, False, True),  # round 5: F(4x4,2x2) both directions, 3 channel blocks forward, 24 tiles per image (ragged 32-tile blocks), bias
    (3, 3, 128, 128, 64, 4, 2, 1, False, False),  # round 6: D's first layer at full size on the LDS-halo MFMA kernel, stride 2 (rgbin_conv_kernel<4, 4, 3, 2>)
    (2, 3, 70, 90, 128, 4, 2, 1, False, True),    # same kernel: ragged 16 x 32 output tiles (35 x 45 map), two channel blocks, bias
    (3, 3, 128, 128, 64, 7, 2, 1, False, True),   # round 6: E's first layer at full size (7x7 / stride 2 / pad 1 -> 62 x 62), bias (rgbin_conv_kernel<7, 7, 3, 2>)
    (2, 3, 71, 77, 64, 7, 2, 1, False, False),    # same kernel, odd sizes (33 x 36 map)
    (40, 64, 16, 16, 64, 4, 2, 1, False, False),  # round 5: F(4x4,2x2) transposed form on 160 tiles x 4 phases = 20 items over persistent workgroups
]


@pytest.fixture(params=["default-dispatch", "winograd-forced"])
def dispatch(request):
    """Every conv case runs through the default kernel choice AND with the size thresholds of the Winograd kernels
    switched off (SRGAN_WINOGRAD_THRESHOLD_SCALE=0, read per call), so that small shapes exercise those kernels too."""
    import os
    if request.param == "winograd-forced":
        os.environ["SRGAN_WINOGRAD_THRESHOLD_SCALE"] = "0"
    yield request.param
    os.environ.pop("SRGAN_WINOGRAD_THRESHOLD_SCALE", None)


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_fwd_bwd(ops, case, dispatch):
    n, i, h, w, o, k, s, p, reflect, has_bias = case
    torch.set_num_threads(16)
    x = rnd(n, i, h, w, seed=1).requires_grad_(True)
    wt = (rnd(o, i, k, k, seed=2) / np.sqrt(i * k * k)).requires_grad_(True)
    b = (rnd(o, seed=3) * 0.1).requires_grad_(True) if has_bias else None
    if reflect:
        yr = F.conv2d(F.pad(x, (p, p, p, p), mode="reflect"), wt, b, s, 0)
    else:
        yr = F.conv2d(x, wt, b, s, p)
    gy = rnd(*yr.shape, seed=4)
    yr.backward(gy)

    xd = x.detach().cuda().requires_grad_(True)
    wd = wt.detach().cuda().requires_grad_(True)
    bd = b.detach().cuda().requires_grad_(True) if has_bias else None
    y = ops.conv2d(xd, wd, bd, s, p, ops.PAD_REFLECT if reflect else ops.PAD_ZERO)
    assert ops.is_nhwc_dense(y)
    y.backward(gy.cuda())
    close(y, yr)
    close(xd.grad, x.grad)
    close(wd.grad, wt.grad, 5e-5)
    if has_bias:
        close(bd.grad, b.grad, 5e-5)


def _random_conv_cases(n_cases, seed):
    """Seeded random layer geometries across every dispatch family (narrow / RGB / stride-2 phases / split-K / Winograd / GEMM)."""
    rng = np.random.RandomState(seed)
    cases = []
    while len(cases) < n_cases:
        k = int(rng.choice([1, 3, 3, 4, 4, 5, 7]))
        s_ = int(rng.choice([1, 1, 2]))
        i = int(rng.choice([3, 4, 16, 32, 48, 64, 96, 128, 256]))
        o = int(rng.choice([1, 3, 4, 16, 32, 64, 96, 128, 512]))
        h, w = int(rng.randint(max(k, 3), 41)), int(rng.randint(max(k, 3), 41))
        p_ = int(rng.randint(0, k // 2 + 1))
        reflect = bool(s_ == 1 and 0 < p_ < min(h, w) and k == 3 and rng.rand() < 0.4)
        n = int(rng.randint(1, 7))
        if (h + 2 * p_ - k) // s_ + 1 < 1 or (w + 2 * p_ - k) // s_ + 1 < 1 or n * i * h * w > 3_000_000:
            continue
        cases.append((n, i, h, w, o, k, s_, p_, reflect, bool(rng.rand() < 0.5)))
    return cases


@pytest.mark.parametrize("case", _random_conv_cases(48, seed=20260410))
def test_conv2d_random_geometries(ops, case, dispatch):
    """The same check as test_conv2d_fwd_bwd on seeded random geometries, inside a packed-weight scope (the trainer's mode: packed
    operands, split-K slabs, kept V images) -- corners of the dispatch that the hand-picked cases may miss."""
    n, i, h, w, o, k, s, p, reflect, has_bias = case
    torch.set_num_threads(16)
    x = rnd(n, i, h, w, seed=1).requires_grad_(True)
    wt = (rnd(o, i, k, k, seed=2) / np.sqrt(i * k * k)).requires_grad_(True)
    b = (rnd(o, seed=3) * 0.1).requires_grad_(True) if has_bias else None
    yr = F.conv2d(F.pad(x, (p, p, p, p), mode="reflect"), wt, b, s, 0) if reflect else F.conv2d(x, wt, b, s, p)
    gy = rnd(*yr.shape, seed=4)
    yr.backward(gy)
    xd, wd = x.detach().cuda().requires_grad_(True), wt.detach().cuda().requires_grad_(True)
    bd = b.detach().cuda().requires_grad_(True) if has_bias else None
    ops.invalidate_packed()
    with ops.pack_cache():
        y = ops.conv2d(xd, wd, bd, s, p, ops.PAD_REFLECT if reflect else ops.PAD_ZERO)
        y.backward(gy.cuda())
    ops.invalidate_packed()
    close(y, yr)
    close(xd.grad, x.grad)
    close(wd.grad, wt.grad, 5e-5)
    if has_bias:
        close(bd.grad, b.grad, 5e-5)


@pytest.mark.parametrize("case", [(32, 64, 32, 32, 128, True), (3, 64, 8, 12, 96, True), (5, 128, 16, 16, 64, False),
                                  (2, 256, 32, 32, 256, False)])
def test_conv2d_f43_weight_gradient_from_kept_image(ops, case, dispatch):
    """Inside a pack-cache scope (the trainer's mode) the forward of an F(4x4,3x3) layer keeps its transformed input and the
    weight gradient is computed from it (srgan_conv2d_wgrad_v): ragged tile blocks, several splits, bias."""
    n, i, h, w, o, has_bias = case
    torch.set_num_threads(16)
    x = rnd(n, i, h, w, seed=11).requires_grad_(True)
    wt = (rnd(o, i, 3, 3, seed=12) / np.sqrt(i * 9)).requires_grad_(True)
    b = (rnd(o, seed=13) * 0.1).requires_grad_(True) if has_bias else None
    yr = F.conv2d(x, wt, b, 1, 1)
    gy = rnd(*yr.shape, seed=14)
    yr.backward(gy)
    xd = x.detach().cuda().requires_grad_(True)
    wd = wt.detach().cuda().requires_grad_(True)
    bd = b.detach().cuda().requires_grad_(True) if has_bias else None
    ops.invalidate_packed()            # entries of earlier tests were laid out under another dispatch setting
    with ops.pack_cache():
        y = ops.conv2d(xd, wd, bd, 1, 1, ops.PAD_ZERO)
        y.backward(gy.cuda())
    ops.invalidate_packed()
    close(y, yr)
    close(xd.grad, x.grad)
    close(wd.grad, wt.grad, 5e-5)
    if has_bias:
        close(bd.grad, b.grad, 5e-5)


@pytest.mark.parametrize("case", [(32, 64, 32, 32, 64), (3, 64, 8, 12, 64), (2, 16, 9, 7, 16), (2, 256, 32, 32, 256)])
@pytest.mark.parametrize("packed", [True, False])
def test_conv2d_skip_gradient_rides_in_the_dgrad_epilogue(ops, case, packed, dispatch):
    """conv2d_skip returns (conv(x), x); the gradient of the second result (the residual connection of SingleResidualBlock,
    model.py:196-201) is added by the input-gradient kernel -- the F(4x4,3x3) epilogue, or one in-place pass on the other
    dispatches -- with and without the packed-weight scope."""
    n, i, h, w, o = case
    torch.set_num_threads(16)
    x = rnd(n, i, h, w, seed=21).requires_grad_(True)
    wt = (rnd(o, i, 3, 3, seed=22) / np.sqrt(i * 9)).requires_grad_(True)
    yr = F.conv2d(x, wt, None, 1, 1)
    gy, gs = rnd(*yr.shape, seed=23), rnd(n, i, h, w, seed=24)
    ((yr * gy).sum() + (x * gs).sum()).backward()
    xd = x.detach().cuda().requires_grad_(True)
    wd = wt.detach().cuda().requires_grad_(True)
    ops.invalidate_packed()
    import contextlib
    with (ops.pack_cache() if packed else contextlib.nullcontext()):
        y, skip = ops.conv2d_skip(xd, wd, None, 1, 1, ops.PAD_ZERO)
        assert skip.shape == xd.shape and torch.equal(skip, xd)
        ((y * gy.cuda()).sum() + (skip * gs.cuda()).sum()).backward()
    ops.invalidate_packed()
    close(y, yr)
    close(xd.grad, x.grad)
    close(wd.grad, wt.grad, 5e-5)
    # only the skip path used: the gradient passes straight through
    xe = x.detach().cuda().requires_grad_(True)
    _, skip = ops.conv2d_skip(xe, wd.detach(), None, 1, 1, ops.PAD_ZERO)
    (skip * gs.cuda()).sum().backward()
    close(xe.grad, gs)


@pytest.mark.parametrize("n,c,o,affine", [(32, 256, 256, True), (4, 64, 64, True), (3, 64, 96, False)])
def test_instance_norm_act_conv_equals_the_unfused_chain(ops, n, c, o, affine):
    """instance_norm_act_conv (norm + ReLU written straight as the F(4x4,3x3) V image, multiply, weight gradient from V) against
    the chain it replaces on the HIP path and against PyTorch-CPU."""
    import os
    torch.set_num_threads(16)
    x = rnd(n, c, 32, 32, seed=31)
    wt = rnd(o, c, 3, 3, seed=32) / np.sqrt(c * 9)
    sc = (torch.rand(n, c, generator=torch.Generator().manual_seed(33)) + 0.5) if affine else None
    sh = rnd(n, c, seed=34) * 0.3 if affine else None
    gy = rnd(n, o, 32, 32, seed=35)
    res = {}
    os.environ["SRGAN_WINOGRAD_THRESHOLD_SCALE"] = "0"          # small batches: force the F(4x4,3x3) dispatch
    try:
        for mode in ("fused", "chain"):
            xd, wd = x.cuda().requires_grad_(True), wt.cuda().requires_grad_(True)
            scd = sc.cuda().requires_grad_(True) if affine else None
            shd = sh.cuda().requires_grad_(True) if affine else None
            ops.invalidate_packed()
            with ops.pack_cache():
                if mode == "fused":
                    assert ops.norm_act_conv_fusable(xd, wd)
                    y = ops.instance_norm_act_conv(xd, scd, shd, wd, ops.ACT_RELU, 0.0, 1e-5)
                else:
                    y = ops.conv2d(ops.instance_norm_act(xd, scd, shd, None, ops.ACT_RELU, 0.0, 1e-5), wd, None, 1, 1)
                y.backward(gy.cuda())
            res[mode] = (y.detach(), xd.grad, wd.grad, scd.grad if affine else None, shd.grad if affine else None)
    finally:
        os.environ.pop("SRGAN_WINOGRAD_THRESHOLD_SCALE", None)
        ops.invalidate_packed()
    # same statistics and expression; the V image is transformed from LDS instead of from memory (the compiler contracts the
    # transform's multiply-adds differently): equal to fp32 rounding, 5e-6 of the scale measured
    close(res["fused"][0], res["chain"][0], 2e-5)
    for a, b in zip(res["fused"][1:], res["chain"][1:]):
        if a is not None:
            close(a, b, 5e-5)
    # and against the PyTorch-CPU reference of the chain
    xr, wr = x.clone().requires_grad_(True), wt.clone().requires_grad_(True)
    hr = F.instance_norm(xr, eps=1e-5)
    if affine:
        scr, shr = sc.clone().requires_grad_(True), sh.clone().requires_grad_(True)
        hr = hr * scr[:, :, None, None] + shr[:, :, None, None]
    yr = F.conv2d(torch.relu(hr), wr, None, 1, 1)
    yr.backward(gy)
    close(res["fused"][0], yr, 1e-4)
    close(res["fused"][1], xr.grad, 2e-4)
    close(res["fused"][2], wr.grad, 1e-4)
    if affine:
        close(res["fused"][3], scr.grad, 2e-4)
        close(res["fused"][4], shr.grad, 2e-4)


@pytest.mark.parametrize("n,c", [(32, 256), (4, 64), (3, 128)])
def test_residual_block_fused_node_vs_chain_and_cpu(ops, n, c):
    """ops.residual_block (one autograd node: V images in the forward, norm backwards writing the input-gradient / weight-gradient
    transforms, skip gradient in the epilogue) against the chain of separate ops on the HIP path and against PyTorch-CPU:
    output and all seven gradients."""
    import os
    torch.set_num_threads(16)
    g = torch.Generator().manual_seed(41)
    x = rnd(n, c, 32, 32, seed=41)
    w1, w2 = rnd(c, c, 3, 3, seed=42) / np.sqrt(c * 9), rnd(c, c, 3, 3, seed=43) / np.sqrt(c * 9)
    s1, s2 = torch.rand(n, c, generator=g) + 0.5, torch.rand(n, c, generator=g) + 0.5
    h1, h2 = rnd(n, c, seed=44) * 0.3, rnd(n, c, seed=45) * 0.3
    gy = rnd(n, c, 32, 32, seed=46)
    os.environ["SRGAN_WINOGRAD_THRESHOLD_SCALE"] = "0"
    res = {}
    try:
        for mode in ("fused", "chain"):
            t = [v.cuda().requires_grad_(True) for v in (x, s1, h1, s2, h2, w1, w2)]
            ops.invalidate_packed()
            with ops.pack_cache():
                if mode == "fused":
                    assert ops.res_block_fusable(t[0], t[5], t[6], t[1], t[3])
                    y = ops.residual_block(*t)
                else:
                    y1 = ops.conv2d(t[0], t[5], None, 1, 1)
                    hh = ops.instance_norm_act(y1, t[1], t[2], None, ops.ACT_RELU)
                    y2 = ops.conv2d(hh, t[6], None, 1, 1)
                    y = ops.instance_norm_act(y2, t[3], t[4], t[0], ops.ACT_NONE)
                y.backward(gy.cuda())
            res[mode] = [y.detach()] + [v.grad for v in t]
    finally:
        os.environ.pop("SRGAN_WINOGRAD_THRESHOLD_SCALE", None)
        ops.invalidate_packed()
    for a, b in zip(res["fused"], res["chain"]):
        close(a, b, 5e-5)
    r = [v.clone().requires_grad_(True) for v in (x, s1, h1, s2, h2, w1, w2)]
    y1 = F.conv2d(r[0], r[5], None, 1, 1)
    hh = torch.relu(F.instance_norm(y1, eps=1e-5) * r[1][:, :, None, None] + r[2][:, :, None, None])
    y2 = F.conv2d(hh, r[6], None, 1, 1)
    yr = F.instance_norm(y2, eps=1e-5) * r[3][:, :, None, None] + r[4][:, :, None, None] + r[0]
    yr.backward(gy)
    close(res["fused"][0], yr, 1e-4)
    # 8.4 M ReLU inputs at the largest size: a few lie within fp32 rounding of zero and take the other branch on the CPU (the HIP
    # chain above, which shares the forward's rounding, agrees to 5e-5) -- flip-tolerant comparison, tests/common.py
    from tests.common import close_grad
    for name, a, b in zip(("dx", "ds1", "dh1", "ds2", "dh2", "dw1", "dw2"), res["fused"][1:], r):
        close_grad(a, b.grad, 3e-4, what=name)


@pytest.mark.parametrize("n,c,hw", [(32, 256, 32), (64, 64, 32), (32, 128, 32), (8, 256, 64)])
def test_residual_block_bf16_storage(ops, n, c, hw):
    """bf16 mode: the residual block as one node with bf16 INTERMEDIATES in HBM (ops._ResBlockBf16Fn: conv outputs, normalised
    activation, conv-output gradients stored as bf16; residual stream, statistics, weight gradients fp32) against the unfused
    bf16-mode chain on the HIP path (fp32 tensors, operands rounded on the way into the MFMA) and against PyTorch-CPU fp32.
    Bounds are relative L2 errors: the fused node adds one bf16 rounding (2^-9) per stored tensor to the chain's.  hw = 64 (the
    trunk maps of configs[4], 256 x 256 images) is past the slab norm kernels: the node's norms are then the two-pass kernels
    with 16-bit I/O (srgan_instnorm_fwd_io / _bwd_io)."""
    import os
    torch.set_num_threads(16)
    x = rnd(n, c, hw, hw, seed=51)
    w1, w2 = rnd(c, c, 3, 3, seed=52) / np.sqrt(c * 9), rnd(c, c, 3, 3, seed=53) / np.sqrt(c * 9)
    g = torch.Generator().manual_seed(54)
    s1, s2 = torch.rand(n, c, generator=g) + 0.5, torch.rand(n, c, generator=g) + 0.5
    h1, h2 = rnd(n, c, seed=55) * 0.3, rnd(n, c, seed=56) * 0.3
    gy = rnd(n, c, hw, hw, seed=57)

    def rel(a, b):
        a, b = a.detach().double().cpu(), b.detach().double().cpu()
        return float((a - b).norm() / b.norm())

    res = {}
    ops.set_compute_dtype("bf16")
    try:
        for mode in ("fused", "chain"):
            t = [v.clone().cuda().requires_grad_(True) for v in (x, s1, h1, s2, h2, w1, w2)]
            ops.RESBLOCK_BF16_STORAGE = mode != "chain"
            try:
                with ops.pack_cache():
                    if mode == "fused":
                        assert ops.res_block_bf16_fusable(t[0], t[5], t[6], t[1], t[3])
                        y = ops.residual_block_bf16(*t)
                    else:
                        assert not ops.res_block_bf16_fusable(t[0], t[5], t[6], t[1], t[3])
                        y1, skip = ops.conv2d_skip(t[0], t[5], None, 1, 1, ops.PAD_ZERO)
                        hh = ops.instance_norm_act(y1, t[1], t[2], None, ops.ACT_RELU)
                        y2 = ops.conv2d(hh, t[6], None, 1, 1)
                        y = ops.instance_norm_act(y2, t[3], t[4], skip, ops.ACT_NONE)
                    y.backward(gy.cuda())
            finally:
                ops.RESBLOCK_BF16_STORAGE = True
            res[mode] = [y.detach()] + [v.grad for v in t]
    finally:
        ops.set_compute_dtype("fp32")
        ops.invalidate_packed()
    r = [v.clone().requires_grad_(True) for v in (x, s1, h1, s2, h2, w1, w2)]
    y1 = F.conv2d(r[0], r[5], None, 1, 1)
    hh = torch.relu(F.instance_norm(y1, eps=1e-5) * r[1][:, :, None, None] + r[2][:, :, None, None])
    y2 = F.conv2d(hh, r[6], None, 1, 1)
    yr = F.instance_norm(y2, eps=1e-5) * r[3][:, :, None, None] + r[4][:, :, None, None] + r[0]
    yr.backward(gy)
    ref = [yr] + [v.grad for v in r]
    names = ("out", "dx", "ds1", "dh1", "ds2", "dh2", "dw1", "dw2")
    log = os.environ.get("SRGAN_TEST_LOG")
    worst = []
    for name, a, b, f in zip(names, res["fused"], res["chain"], ref):
        e_chain, e_ref, e_chain_ref = rel(a, b), rel(a, f), rel(b, f)
        if log:
            print(f"resblock16 n={n} c={c} hw={hw} {name:4s} fused-vs-chain {e_chain:.2e}  fused-vs-fp32 {e_ref:.2e}  chain-vs-fp32 {e_chain_ref:.2e}")
        worst.append((name, e_chain, e_ref, e_chain_ref))
    for name, e_chain, e_ref, e_chain_ref in worst:
        # dx / ds1 / dh1 / dw1 carry the ReLU-mask flips of bf16-rounded conv outputs (3e-2 .. 5e-2 of the fp32 gradient on BOTH bf16
        # paths; measured at n=32, c=256: fused-vs-fp32 within 2 % of chain-vs-fp32 on every tensor, fused-vs-chain <= 0.45 of
        # chain-vs-fp32): the fused node is held to the chain's own distance from fp32, and to the chain itself
        assert e_chain <= 0.6 * e_chain_ref + 5e-3, (name, e_chain, e_chain_ref)
        assert e_ref <= 1.15 * e_chain_ref + 3e-3, (name, e_ref, e_chain_ref)


def test_packed_cache_releases_dead_networks(ops):
    """The packed-operand cache holds parameters weakly: operands of a network that no longer exists are dropped at the next
    scope entry (they used to pin the weight, its packed buffer and a slot of every later refresh for the life of the process)."""
    import gc
    ops.invalidate_packed()
    x = rnd(2, 32, 12, 12, seed=1).cuda()

    def use(seed):
        w = (rnd(64, 32, 3, 3, seed=seed) / 17).cuda().requires_grad_(True)
        with ops.pack_cache():
            ops.conv2d(x, w, None, 1, 1).sum().backward()
        return w

    w_live = use(2)
    n_live = len(ops._pack_cache)
    assert n_live >= 1                                   # the forward operand (x needs no gradient, so no input-gradient operand)
    w_dead = use(3)
    assert len(ops._pack_cache) == 2 * n_live
    del w_dead
    gc.collect()
    with ops.pack_cache():
        assert len(ops._pack_cache) == n_live and all(h.weight is w_live for h in ops._pack_cache.values())
    ops.invalidate_packed()


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_implicit_gemm_operands_are_shared_between_batch_sizes(ops, mode):
    """Round 6: srgan_conv2d_pack_signature also covers the implicit-GEMM operands -- a hash of every layout field of the pack --
    so a weight used at two batch sizes (the discriminators: 64 images in their own update, 32 in the generator's; the encoder:
    32 and 64) keeps ONE operand per direction instead of one per geometry, and the per-optimiser-step repack writes it once.
    An encoder-like 3x3 reflect layer on a 7 x 7 map and a discriminator-like 4x4 / stride-2 layer on a 16 x 16 map, batch 4 and 8:
    two cached operands per weight (forward, input gradient), and the same bits as calls outside any cache."""
    torch.manual_seed(11)
    layers = [((rnd(512, 512, 3, 3, seed=1) / 68).cuda(), 7, 1, 1, True), ((rnd(512, 256, 4, 4, seed=2) / 64).cuda(), 16, 2, 1, False)]
    ops.set_compute_dtype(mode)
    try:
        for wt, hw, stride, pad, reflect in layers:
            pm = ops.PAD_REFLECT if reflect else ops.PAD_ZERO
            xs = [rnd(n, wt.shape[1], hw, hw, seed=3 + n).cuda() for n in (4, 8)]

            def run(x):
                xv, wv = x.clone().requires_grad_(True), wt.clone().requires_grad_(True)
                y = ops.conv2d(xv, wv, None, stride, pad, pm)
                y.backward(torch.ones_like(y))
                return y.detach(), xv.grad, wv.grad

            plain = [run(x) for x in xs]
            with ops.pack_cache():
                wv = wt.clone().requires_grad_(True)
                before = len(ops._pack_cache)
                got = []
                for x in xs:
                    xv = x.clone().requires_grad_(True)
                    wv.grad = None
                    y = ops.conv2d(xv, wv, None, stride, pad, pm)
                    y.backward(torch.ones_like(y))
                    got.append((y.detach(), xv.grad, wv.grad.clone()))
                assert len(ops._pack_cache) - before == 2, [k for k in ops._pack_cache][before:]
            for a, b in zip(plain, got):
                for u, v in zip(a, b):
                    assert torch.equal(u, v)
            ops.invalidate_packed()
    finally:
        ops.set_compute_dtype("fp32")
        ops.invalidate_packed()


def _bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


# layers the bf16 kernels serve: vector-gather implicit GEMM (Cin % 32 == 0, Cout >= 32); RGB / 1-channel heads stay fp32
@pytest.mark.parametrize("case", [c for c in CONV_CASES if c[1] % 32 == 0 and c[4] >= 32 and c[0] * c[2] * c[3] <= 70000])
def test_conv2d_bf16_compute_mode(ops, case):
    """bf16 MFMA mode (BASELINE configs [2]-[4]): forward, input gradient and weight gradient multiply bf16-rounded operands
    (x * w, dy * w, dy * x) with fp32 accumulation, so each equals the fp32 convolution of the bf16-ROUNDED operands up to
    summation order -- held at the same 2e-5 as the fp32 kernels."""
    n, i, h, w, o, k, s, p, reflect, has_bias = case
    torch.set_num_threads(16)
    x = rnd(n, i, h, w, seed=1)
    wt = rnd(o, i, k, k, seed=2) / np.sqrt(i * k * k)
    b = (rnd(o, seed=3) * 0.1) if has_bias else None

    def ref(xv, wv, gyv=None):
        xv, wv = xv.clone().requires_grad_(True), wv.clone().requires_grad_(True)
        xin = F.pad(xv, (p, p, p, p), mode="reflect") if reflect else xv
        yv = F.conv2d(xin, wv, b, s, 0 if reflect else p)
        if gyv is not None:
            yv.backward(gyv)
        return yv.detach(), xv.grad, wv.grad

    y_ref, _, _ = ref(_bf16_round(x), _bf16_round(wt))
    gy = rnd(*y_ref.shape, seed=4)
    _, dx_ref, _ = ref(x, _bf16_round(wt), _bf16_round(gy))          # dx = dgrad(bf16(dy), bf16(w))
    _, _, dw_ref = ref(_bf16_round(x), wt, _bf16_round(gy))           # dw = wgrad(bf16(x), bf16(dy))
    ops.set_compute_dtype("bf16")
    try:
        assert ops.get_compute_dtype() == "bf16"
        xd = x.cuda().requires_grad_(True)
        wd = wt.cuda().requires_grad_(True)
        bd = b.cuda().requires_grad_(True) if has_bias else None
        y = ops.conv2d(xd, wd, bd, s, p, ops.PAD_REFLECT if reflect else ops.PAD_ZERO)
        y.backward(gy.cuda())
        close(y, y_ref)
        close(xd.grad, dx_ref)
        close(wd.grad, dw_ref)
    finally:
        ops.set_compute_dtype("fp32")


def _random_bf16_conv_cases(n_cases, seed):
    """Seeded random geometries of the layers the bf16 kernels serve (Cin a multiple of 32, Cout >= 32): patch kernels, the
    64-deep-K implicit GEMM (Cin % 64 == 0: ragged pixel / channel tiles, stride-2 phases, reflect, 1x1, split-K) and the rest."""
    rng = np.random.RandomState(seed)
    cases = []
    while len(cases) < n_cases:
        k = int(rng.choice([1, 3, 3, 4, 4]))
        s_ = int(rng.choice([1, 1, 2]))
        i = int(rng.choice([32, 64, 64, 96, 128, 128, 192, 256]))
        o = int(rng.choice([32, 64, 96, 128, 160, 256, 512]))
        h, w = int(rng.randint(max(k, 3), 41)), int(rng.randint(max(k, 3), 41))
        p_ = int(rng.randint(0, k // 2 + 1))
        reflect = bool(s_ == 1 and 0 < p_ < min(h, w) and k == 3 and rng.rand() < 0.4)
        n = int(rng.randint(1, 9))
        if (h + 2 * p_ - k) // s_ + 1 < 1 or (w + 2 * p_ - k) // s_ + 1 < 1 or n * i * h * w > 3_000_000:
            continue
        cases.append((n, i, h, w, o, k, s_, p_, reflect, bool(rng.rand() < 0.5)))
    return cases


@pytest.mark.parametrize("case", _random_bf16_conv_cases(32, seed=20261004))
def test_conv2d_bf16_random_geometries(ops, case):
    """test_conv2d_bf16_compute_mode's check on seeded random geometries inside a packed-weight scope (the trainer's mode: bf16
    packed operands of igemm16_kernel, split-K slabs, persistent tile loop)."""
    n, i, h, w, o, k, s, p, reflect, has_bias = case
    torch.set_num_threads(16)
    x = rnd(n, i, h, w, seed=1)
    wt = rnd(o, i, k, k, seed=2) / np.sqrt(i * k * k)
    b = (rnd(o, seed=3) * 0.1) if has_bias else None

    def ref(xv, wv, gyv=None):
        xv, wv = xv.clone().requires_grad_(True), wv.clone().requires_grad_(True)
        xin = F.pad(xv, (p, p, p, p), mode="reflect") if reflect else xv
        yv = F.conv2d(xin, wv, b, s, 0 if reflect else p)
        if gyv is not None:
            yv.backward(gyv)
        return yv.detach(), xv.grad, wv.grad

    y_ref, _, _ = ref(_bf16_round(x), _bf16_round(wt))
    gy = rnd(*y_ref.shape, seed=4)
    _, dx_ref, _ = ref(x, _bf16_round(wt), _bf16_round(gy))
    _, _, dw_ref = ref(_bf16_round(x), wt, _bf16_round(gy))
    ops.set_compute_dtype("bf16")
    try:
        xd, wd = x.cuda().requires_grad_(True), wt.cuda().requires_grad_(True)
        bd = b.cuda().requires_grad_(True) if has_bias else None
        ops.invalidate_packed()
        with ops.pack_cache():
            y = ops.conv2d(xd, wd, bd, s, p, ops.PAD_REFLECT if reflect else ops.PAD_ZERO)
            y.backward(gy.cuda())
        try:
            close(y, y_ref)
        except AssertionError:
            # heads with a 1 x 1 output map run on the exact-fp32 direct kernels in both modes (csrc/conv_narrow.hip)
            close(y, ref(x, wt)[0])
        close(xd.grad, dx_ref)
        close(wd.grad, dw_ref, 5e-5)
    finally:
        ops.set_compute_dtype("fp32")
        ops.invalidate_packed()


@pytest.mark.parametrize("case", [(3, 64, 8, 32, 64), (2, 256, 32, 32, 256), (1, 128, 12, 64, 128)])
def test_conv2d_skip_bf16_mode(ops, case):
    """bf16 mode on the residual-trunk shapes: the skip path's gradient is added in the epilogue of the LDS-resident-patch kernel
    (conv_halo16.hip) -- same bf16-rounded-operand references as test_conv2d_bf16_compute_mode, plus the fp32 skip gradient."""
    n, i, h, w, o = case
    torch.set_num_threads(16)
    x = rnd(n, i, h, w, seed=41)
    wt = rnd(o, i, 3, 3, seed=42) / np.sqrt(i * 9)
    gy, gs = rnd(n, o, h, w, seed=43), rnd(n, i, h, w, seed=44)
    xr = x.clone().requires_grad_(True)
    yr = F.conv2d(_bf16_round(x), _bf16_round(wt), None, 1, 1)
    (F.conv2d(xr, _bf16_round(wt), None, 1, 1) * _bf16_round(gy)).sum().backward()
    dx_ref = xr.grad + gs
    ops.set_compute_dtype("bf16")
    try:
        xd, wd = x.cuda().requires_grad_(True), wt.cuda().requires_grad_(True)
        with ops.pack_cache():
            y, skip = ops.conv2d_skip(xd, wd, None, 1, 1, ops.PAD_ZERO)
            ((y * gy.cuda()).sum() + (skip * gs.cuda()).sum()).backward()
        close(y, yr)
        close(xd.grad, dx_ref)
    finally:
        ops.set_compute_dtype("fp32")
        ops.invalidate_packed()


def test_conv2d_fused_leaky_relu(ops, dispatch):
    x = rnd(2, 32, 10, 10, seed=1).requires_grad_(True)
    wt = (rnd(64, 32, 4, 4, seed=2) / 20).requires_grad_(True)
    yr = F.leaky_relu(F.conv2d(x, wt, None, 2, 1), 0.01)
    gy = rnd(*yr.shape, seed=4)
    yr.backward(gy)
    xd, wd = x.detach().cuda().requires_grad_(True), wt.detach().cuda().requires_grad_(True)
    y = ops.conv2d(xd, wd, None, 2, 1, ops.PAD_ZERO, ops.ACT_LRELU, 0.01)
    y.backward(gy.cuda())
    close(y, yr)
    close(xd.grad, x.grad)
    close(wd.grad, wt.grad, 5e-5)


@pytest.mark.parametrize("case", [(2, 64, 8, 8, 32), (2, 8, 5, 5, 4), (1, 256, 4, 4, 128)])
def test_conv_transpose2d(ops, case, dispatch):
    n, ci, h, w, co = case
    x = rnd(n, ci, h, w, seed=1).requires_grad_(True)
    wt = (rnd(ci, co, 4, 4, seed=2) / np.sqrt(ci * 4)).requires_grad_(True)
    yr = F.conv_transpose2d(x, wt, None, 2, 1)
    gy = rnd(*yr.shape, seed=4)
    yr.backward(gy)
    xd, wd = x.detach().cuda().requires_grad_(True), wt.detach().cuda().requires_grad_(True)
    y = ops.conv_transpose2d(xd, wd, 2, 1)
    y.backward(gy.cuda())
    close(y, yr)
    close(xd.grad, x.grad)
    close(wd.grad, wt.grad, 5e-5)


def test_conv_weight_read_at_backward_time(ops):
    """torch-1.4 stale-graph semantics: dgrad uses the weights as they are when backward runs."""
    x = rnd(1, 32, 6, 6, seed=1).cuda().requires_grad_(True)
    w = (rnd(32, 32, 3, 3, seed=2) / 17).cuda().requires_grad_(True)
    y = ops.conv2d(x, w, None, 1, 1)
    w_new = rnd(32, 32, 3, 3, seed=9) / 17
    w.data.copy_(w_new.cuda())            # what the fused Adam does through raw pointers
    gy = rnd(*y.shape, seed=3)
    y.backward(gy.cuda())
    ref = torch.nn.grad.conv2d_input(x.shape, w_new, gy, 1, 1)
    close(x.grad, ref)


@pytest.mark.parametrize("shape,act", [((2, 16, 9, 9), 1), ((3, 64, 16, 16), 0), ((2, 8, 31, 31), 2), ((2, 6, 5, 5), 1),
                                       ((8, 512, 16, 16), 1), ((16, 256, 7, 9), 2), ((16, 256, 32, 32), 1),   # single-pass slab kernels
                                       ((64, 64, 20, 20), 0)])
@pytest.mark.parametrize("affine", [False, True])
def test_instance_norm_act(ops, shape, act, affine):
    n, c, h, w = shape
    x = (rnd(*shape, seed=1) * 2 + 0.5).requires_grad_(True)
    res = rnd(*shape, seed=5).requires_grad_(True)
    sc = (1 + 0.3 * rnd(n, c, seed=2)).requires_grad_(True) if affine else None
    sh = (0.5 * rnd(n, c, seed=3)).requires_grad_(True) if affine else None
    xh = F.instance_norm(x, eps=1e-5)
    z = xh * sc[:, :, None, None] + sh[:, :, None, None] if affine else xh
    z = {0: z, 1: torch.relu(z), 2: F.leaky_relu(z, 0.2)}[act]
    yr = z + res
    gy = rnd(*shape, seed=4)
    yr.backward(gy)
    xd = x.detach().cuda().requires_grad_(True)
    rd = res.detach().cuda().requires_grad_(True)
    scd = sc.detach().cuda().requires_grad_(True) if affine else None
    shd = sh.detach().cuda().requires_grad_(True) if affine else None
    y = ops.instance_norm_act(xd, scd, shd, rd, act, 0.2)
    y.backward(gy.cuda())
    close(y, yr, 2e-5)
    close(xd.grad, x.grad, 1e-4)
    close(rd.grad, res.grad)
    if affine:
        close(scd.grad, sc.grad, 1e-4)
        close(shd.grad, sh.grad, 1e-4)


def test_cbin_affine(ops):
    n, ch, nc = 5, 24, 12
    c = rnd(n, nc, seed=1).requires_grad_(True)
    W = (rnd(ch, nc, seed=2) * 0.3).requires_grad_(True)
    b = (rnd(ch, seed=3) * 0.1).requires_grad_(True)
    gam = (1 + 0.2 * rnd(ch, seed=4)).requires_grad_(True)
    bet = (0.1 * rnd(ch, seed=5)).requires_grad_(True)
    t = torch.tanh(F.linear(c, W, b))
    scale_r = gam[None, :].expand(n, ch)
    shift_r = t * gam + bet
    g1, g2 = rnd(n, ch, seed=6), rnd(n, ch, seed=7)
    ((scale_r * g1).sum() + (shift_r * g2).sum()).backward()
    dev = [v.detach().cuda().requires_grad_(True) for v in (c, W, b, gam, bet)]
    scale, shift = ops.cbin_affine(*dev)
    ((scale * g1.cuda()).sum() + (shift * g2.cuda()).sum()).backward()
    close(scale, scale_r)
    close(shift, shift_r)
    for d, r in zip(dev, (c, W, b, gam, bet)):
        close(d.grad, r.grad, 5e-5)


def test_cbin_affine_multi(ops):
    """All central-biasing layers of a network in one launch (forward) / two (backward): three layers of different widths,
    one of them without a gradient on its scale output, against the torch formula; dc sums the layers."""
    n, nc, chs = 7, 12, (24, 64, 256)
    c = rnd(n, nc, seed=1).requires_grad_(True)
    params, outs_r, weights = [], [], []
    for l, ch in enumerate(chs):
        W = (rnd(ch, nc, seed=10 + l) * 0.3).requires_grad_(True)
        b = (rnd(ch, seed=20 + l) * 0.1).requires_grad_(True)
        gam = (1 + 0.2 * rnd(ch, seed=30 + l)).requires_grad_(True)
        bet = (0.1 * rnd(ch, seed=40 + l)).requires_grad_(True)
        params.append((W, b, gam, bet))
        t = torch.tanh(F.linear(c, W, b))
        outs_r.append((gam[None, :].expand(n, ch), t * gam + bet))
        weights.append((rnd(n, ch, seed=50 + l), rnd(n, ch, seed=60 + l)))
    loss = 0
    for l, ((sc, sh), (g1, g2)) in enumerate(zip(outs_r, weights)):
        loss = loss + (sh * g2).sum() + ((sc * g1).sum() if l != 1 else 0)
    loss.backward()
    cd = c.detach().cuda().requires_grad_(True)
    pd = [tuple(v.detach().cuda().requires_grad_(True) for v in p) for p in params]
    outs = ops.cbin_affine_multi(cd, pd)
    loss = 0
    for l, ((sc, sh), (g1, g2)) in enumerate(zip(outs, weights)):
        close(sc, outs_r[l][0])
        close(sh, outs_r[l][1])
        loss = loss + (sh * g2.cuda()).sum() + ((sc * g1.cuda()).sum() if l != 1 else 0)
    loss.backward()
    close(cd.grad, c.grad, 5e-5)
    for p, r in zip(pd, params):
        for d, v in zip(p, r):
            close(d.grad, v.grad, 5e-5)


def test_pools_and_heads(ops):
    x = rnd(2, 8, 13, 13, seed=1).requires_grad_(True)
    gy = None
    for name, ref in (("avgpool3s2", lambda t: F.avg_pool2d(t, 3, 2, 1, count_include_pad=False)),
                      ("avgpool2", lambda t: F.avg_pool2d(t, 2, 2))):
        x.grad = None
        yr = ref(x)
        gy = rnd(*yr.shape, seed=2)
        yr.backward(gy)
        xd = x.detach().cuda().requires_grad_(True)
        y = getattr(ops, name)(xd)
        y.backward(gy.cuda())
        close(y, yr)
        close(xd.grad, x.grad)
    # even sizes too (the D input is 128x128)
    x2 = rnd(1, 3, 16, 16, seed=3)
    close(ops.avgpool3s2(x2.cuda()), F.avg_pool2d(x2, 3, 2, 1, count_include_pad=False))
    # LeakyReLU + global average pool
    x.grad = None
    yr = F.leaky_relu(x, 0.2).mean(dim=(2, 3))
    g = rnd(*yr.shape, seed=4)
    yr.backward(g)
    xd = x.detach().cuda().requires_grad_(True)
    y = ops.lrelu_global_avgpool(xd, 0.2)
    y.backward(g.cuda())
    close(y, yr)
    close(xd.grad, x.grad)
    # linear
    a = rnd(6, 1024, seed=5).requires_grad_(True)
    W = (rnd(8, 1024, seed=6) / 32).requires_grad_(True)
    b = rnd(8, seed=7).requires_grad_(True)
    yr = F.linear(a, W, b)
    g = rnd(*yr.shape, seed=8)
    yr.backward(g)
    dev = [v.detach().cuda().requires_grad_(True) for v in (a, W, b)]
    y = ops.linear(*dev)
    y.backward(g.cuda())
    close(y, yr)
    for d, r in zip(dev, (a, W, b)):
        close(d.grad, r.grad, 5e-5)
    # tanh / activation / add
    t = rnd(2, 3, 8, 8, seed=9).requires_grad_(True)
    yr = torch.tanh(t)
    g = rnd(*yr.shape, seed=10)
    yr.backward(g)
    td = t.detach().cuda().requires_grad_(True)
    y = ops.tanh(td)
    y.backward(g.cuda())
    close(y, yr)
    close(td.grad, t.grad)
    u, v = rnd(2, 4, 5, 5, seed=11), rnd(2, 4, 5, 5, seed=12)
    close(ops.add(u.cuda(), v.cuda()), u + v)
    close(ops.activation(u.cuda(), ops.ACT_LRELU, 0.2), F.leaky_relu(u, 0.2))


def test_layout_round_trip(ops):
    x = rnd(3, 5, 7, 9, seed=1)
    xd = ops.to_nhwc(x.cuda())
    assert ops.is_nhwc_dense(xd)
    close(xd, x, 0, 0)
    close(ops.to_nchw(xd), x, 0, 0)
    assert ops.to_nchw(xd).is_contiguous()
    with pytest.raises(ValueError, match="expected 4D input"):
        ops.to_nhwc(torch.zeros(3, 4).cuda())


def test_losses(ops, golden_dir):
    import os
    gold = np.load(os.path.join(golden_dir, "losses.npz"))
    # LSGAN / class MSE against the reference's own numbers
    o1, o2 = torch.from_numpy(gold["ls_o1"]), torch.from_numpy(gold["ls_o2"])
    for tgt, idx in ((1.0, 0), (0.0, 1)):
        v = ops.mse_const(o1.cuda(), tgt, 0.5) + ops.mse_const(o2.cuda(), tgt, 0.5)
        assert abs(float(v) - gold["ls_vals"][idx]) < 1e-5 * max(1, abs(gold["ls_vals"][idx]))
    od = o1.cuda().requires_grad_(True)
    ops.mse_const(od, 1.0, 0.5).backward()
    close(od.grad, 0.5 * 2 * (o1 - 1.0) / o1.numel())
    # softmax + MSE: logits whose softmax is the golden q
    z = torch.from_numpy(gold["ls_q1"]).log().requires_grad_(True)
    lab = torch.from_numpy(gold["ls_lab"])
    q = torch.softmax(z, 1)
    lr = ((q - F.one_hot(lab, 4).float()) ** 2).mean() * 0.7
    lr.backward()
    zd = z.detach().cuda().requires_grad_(True)
    l, qd = ops.softmax_mse(zd, lab, 0.7)
    l.backward()
    close(l, lr, 1e-5)
    close(qd, q, 1e-5)
    close(zd.grad, z.grad, 1e-4)
    # L1
    a, b = rnd(2, 3, 16, 16, seed=1).requires_grad_(True), rnd(2, 3, 16, 16, seed=2).requires_grad_(True)
    lr = (a - b).abs().mean() * 5.0
    lr.backward()
    ad, bd = a.detach().cuda().requires_grad_(True), b.detach().cuda().requires_grad_(True)
    l = ops.l1_mean(ad, bd, 5.0)
    l.backward()
    close(l, lr, 1e-5)
    close(ad.grad, a.grad)
    close(bd.grad, b.grad)
    # latent losses vs the reference's values and autograd gradients
    tgt = torch.from_numpy(gold["hist_target_seed0"]).cuda()
    for name in ("randn1234", "sin"):
        mu = torch.from_numpy(gold[f"{name}_mu"]).cuda().requires_grad_(True)
        total, parts, corr = ops.latent_losses(mu, 32, tgt, 10.0, 100.0, 100.0)
        total.backward()
        ref = gold[f"{name}_vals"]
        np.testing.assert_allclose(parts.cpu().numpy(), ref, rtol=2e-4)
        assert abs(float(total) - (10 * ref[0] + 100 * ref[1] + 100 * ref[2])) < 2e-4 * abs(float(total))
        gref = 10 * gold[f"{name}_dbkl"] + 100 * gold[f"{name}_dcorr"] + 100 * gold[f"{name}_dhist"]
        close(mu.grad, torch.from_numpy(gref), 5e-4)
        close(corr, torch.from_numpy(np.corrcoef(gold[f"{name}_mu"].T)), 1e-5, 1e-6)
    # public util.py-style entry points
    from srgan_amd import losses as hl
    mu = torch.from_numpy(gold["sin_mu"]).cuda()
    close(hl.corrcoef(mu.t()), torch.from_numpy(np.corrcoef(gold["sin_mu"].T)), 1e-5, 1e-6)
    assert abs(float(hl.corrcoef_loss(mu.t(), "cuda")) - gold["sin_vals"][1]) < 1e-5
    torch.manual_seed(0)
    hi = hl.histogram_imitation("cuda")                  # same CPU RNG draw as the reference constructor
    close(hi.target, torch.from_numpy(gold["hist_target_seed0"]), 1e-5)
    assert abs(float(hi.loss(mu)) - gold["sin_vals"][2]) < 2e-4 * gold["sin_vals"][2]
    q = [torch.from_numpy(gold["ls_q1"]).cuda(), torch.from_numpy(gold["ls_q2"]).cuda()]
    oh = hl.class_encode(torch.from_numpy(gold["ls_lab"]), "cuda", np.eye(4))
    assert abs(float(hl.get_domainloss_D(q, oh, torch.nn.MSELoss())) - gold["ls_vals"][2]) < 1e-6
    xs = torch.randn(5000, generator=torch.Generator().manual_seed(3))
    from oracle import losses as ol
    xd = xs.cuda().requires_grad_(True)
    h = hl.GaussianHistogram(50, -10, 10, 0.2)(xd)
    xr = xs.clone().requires_grad_(True)
    hr = ol.soft_histogram(xr)
    w = torch.linspace(0, 1, 50)
    (h * w.cuda()).sum().backward()
    (hr * w).sum().backward()
    close(h, hr, 1e-5)
    close(xd.grad, xr.grad, 1e-4)


def test_cross_entropy(ops):
    z = rnd(9, 4, seed=1).requires_grad_(True)
    lab = torch.tensor([0, 3, 1, 2, 2, 0, 1, 3, 3])
    lr = F.cross_entropy(z, lab)
    lr.backward()
    zd = z.detach().cuda().requires_grad_(True)
    l = ops.softmax_xent(zd, lab.cuda())
    l.backward()
    close(l, lr, 1e-6)
    close(zd.grad, z.grad, 1e-5)


def test_adam_matches_torch14_math(ops):
    from oracle.trainer import Adam14
    p = rnd(1000, seed=1)
    ref = p.clone().requires_grad_(True)
    opt = Adam14([ref], lr=1e-3)
    pd = p.cuda()
    m, v = torch.zeros_like(pd), torch.zeros_like(pd)
    for step in range(1, 4):
        g = rnd(1000, seed=10 + step)
        ref.grad = g.clone()
        opt.step()
        ops.adam_step_(pd, g.cuda(), m, v, 1e-3, 0.5, 0.999, 1e-8, step)
    close(pd, ref, 1e-6)


def test_no_cpu_fallback(ops):
    from srgan_amd._lib import SrganHipError
    with pytest.raises(SrganHipError):
        ops.conv2d(torch.zeros(1, 3, 8, 8), torch.zeros(4, 3, 3, 3))


def test_pack_cache_never_serves_a_stale_operand(ops):
    """The packed-weight cache persists between ``pack_cache`` scopes: version-bumping updates, raw ``.data`` updates made
    outside a scope, swapped storage and an explicit ``refresh_packed`` must all be seen."""
    x = rnd(2, 64, 12, 12, seed=1).cuda()
    w = torch.nn.Parameter((rnd(64, 64, 3, 3, seed=2) / 24).cuda())

    def conv():
        with torch.no_grad():
            return ops.conv2d(x, w, None, 1, 1).clone()

    def ref():
        return F.conv2d(x.cpu(), w.detach().cpu(), None, 1, 1)

    with ops.pack_cache():
        close(conv(), ref())
        with torch.no_grad():
            w.mul_(2.0)                      # version bump inside the scope
        close(conv(), ref())
        w.data.add_(0.01)                    # raw update: only an explicit refresh can see it
        ops.refresh_packed([w])
        close(conv(), ref())
    w.data.mul_(0.5)                         # raw update between scopes: re-packed on entry
    with ops.pack_cache():
        close(conv(), ref())
        w.data = (w.data * 3.0).clone()      # storage swapped under the parameter
        close(conv(), ref())


@pytest.mark.parametrize("mode,defer", [("fp32", "arena"), ("bf16", "arena"), ("fp32", "arena48m"), ("fp32", "immediate")])
def test_fused_param_grads_equal_autograd_accumulation(ops, mode, defer):
    """ops.fused_param_grads: a generator reached through TWO graphs in one backward call (as inside util_notebook.py:664 and
    :689) -- the weight-gradient kernels add the second contribution in their own epilogue (slab reduce with beta = 1,
    central-biasing records with the accumulate flag) instead of autograd's AccumulateGrad input buffer doing it with one
    elementwise launch per parameter.  Same operands, same single addition: every parameter gradient bit-identical; a
    post-accumulate-grad hook still fires once per parameter and backward call (the data-parallel reducer counts on it); a
    gradient already in ``p.grad`` is added to, as AccumulateGrad would.
    ``defer``: inside the scope the split-K slab sums wait in an arena and run as few launches (srgan_wgrad_defer_begin; a second
    sum for the same buffer, a full queue or a full arena launch what is waiting) -- default arena, one so small that it
    overflows and the largest layers do not fit at all, and the immediate sums: the same bits every time."""
    from srgan_amd import model
    saved = (ops._NO_WGRAD_DEFER, ops._WGRAD_ARENA_BYTES)
    arena_keys = lambda: [k for k in ops._workspaces if k[0] == "wgrad_arena"]
    for k in arena_keys():
        del ops._workspaces[k]
    ops._NO_WGRAD_DEFER = defer == "immediate"
    ops._WGRAD_ARENA_BYTES = (48 << 20) if defer == "arena48m" else (1 << 30)      # ("arena": one that holds the whole pass)
    ops._arena_want.clear()

    def defer_stats():
        import ctypes
        from srgan_amd import _lib
        a, b = ctypes.c_longlong(), ctypes.c_longlong()
        _lib.check(_lib.load().srgan_wgrad_defer_stats(ctypes.byref(a), ctypes.byref(b)), "defer_stats")
        return a.value, b.value
    stats0 = defer_stats()
    torch.manual_seed(5)
    G = model.SingleGenerator(3, 64, 2, 2, 2, "instance", num_con=12).cuda()      # full width, two residual blocks
    x1, x2 = rnd(2, 3, 128, 128, seed=1).cuda(), rnd(3, 3, 128, 128, seed=2).cuda()
    c1, c2 = rnd(2, 12, seed=3).cuda(), rnd(3, 12, seed=4).cuda()
    fired = []
    handles = [p.register_post_accumulate_grad_hook(lambda p_: fired.append(id(p_))) for p in G.parameters()]
    ops.set_compute_dtype(mode)
    try:
        out = {}
        for fused in (False, True):
            for p in G.parameters():
                p.grad = None
            fired.clear()
            ops.invalidate_packed()
            with ops.pack_cache():
                y1, y2 = G(x1, c1), G(x2, c2)
                loss = y1.square().mean() + (y2 * 0.5).abs().mean()
                with ops.fused_param_grads(fused):
                    loss.backward(retain_graph=True)
                first = {n: p.grad.clone() for n, p in G.named_parameters()}
                assert sorted(fired) == sorted(id(p) for p in G.parameters())          # once per parameter
                with ops.fused_param_grads(fused):
                    y1.mean().backward()          # a second backward call onto existing .grad tensors
            out[fused] = (first, {n: p.grad.clone() for n, p in G.named_parameters()})
    finally:
        ops.set_compute_dtype("fp32")
        ops.invalidate_packed()
        for h in handles:
            h.remove()
        ops._NO_WGRAD_DEFER, ops._WGRAD_ARENA_BYTES = saved
        sums, launches = (b - a for a, b in zip(stats0, defer_stats()))
        if defer == "immediate":
            assert sums == 0 and not arena_keys()
        elif defer == "arena48m":     # the arena holds two of the trunk's slabs: queued, but launched almost one by one
            assert sums >= 30, (sums, launches)
        else:       # every layer's sum is queued twice per backward call (two uses): a handful of launches
            assert sums >= 30 and launches < sums / 3, (sums, launches)
        for k in arena_keys():
            del ops._workspaces[k]
    for stage in (0, 1):
        for n in out[False][stage]:
            assert torch.equal(out[False][stage][n], out[True][stage][n]), (stage, n)


@pytest.mark.parametrize("o,i", [(256, 256), (64, 96), (128, 64)])
def test_multi_pack_launch_equals_single_pack(ops, o, i):
    """The per-optimiser-step repack (srgan_conv2d_pack_multi: one launch for every cached operand; its F(4x4,3x3) images take the
    workgroup-cooperative path with coalesced 16-byte reads through LDS) must write the same bytes as the per-layer pack kernel
    (srgan_conv2d_pack), forward and input-gradient operands."""
    import os
    w = (rnd(o, i, 3, 3, seed=7) / np.sqrt(i * 9)).cuda().requires_grad_(True)
    x = rnd(2, i, 32, 32, seed=8).cuda().requires_grad_(True)
    os.environ["SRGAN_WINOGRAD_THRESHOLD_SCALE"] = "0"
    try:
        ops.invalidate_packed()
        with ops.pack_cache():
            ops.conv2d(x, w, None, 1, 1).sum().backward()          # packs kind 0 and kind 1 one by one
            hits = [h for h in ops._pack_cache.values() if h.weight is w]
            assert len(hits) == 2
            single = [h.buf.clone() for h in hits]
            for h in hits:
                h.buf.zero_()
            ops.refresh_packed([w], force=True)                     # the multi-pack launch
            for h, s in zip(hits, single):
                assert torch.equal(h.buf, s), (h.kind, o, i)
            # the workgroup-cooperative F(4x4,3x3) path against the one-thread-per-item path (different FMA contraction: compared
            # as floats, to rounding)
            os.environ["SRGAN_PACK_ITEM_PATH"] = "1"
            try:
                ops._tables.clear()
                for h in hits:
                    h.buf.zero_()
                ops.refresh_packed([w], force=True)
            finally:
                os.environ.pop("SRGAN_PACK_ITEM_PATH", None)
                ops._tables.clear()
            for h, s in zip(hits, single):
                close(h.buf.view(torch.float32), s.view(torch.float32), 1e-6)
    finally:
        os.environ.pop("SRGAN_WINOGRAD_THRESHOLD_SCALE", None)
        ops.invalidate_packed()


def test_d_losses_lincomb_kl_normal(ops):
    """The one-launch loss assemblies against plain PyTorch-CPU arithmetic of the reference's formulas: LSGAN + class MSE of a
    real | fake discriminator batch (util.py:457-468 as used by util_notebook.py:582-590), the weighted sum of a phase's loss
    terms, the conventional KL (util_notebook.py:630-634) -- values and every gradient."""
    B, nc = 6, 4
    o1, o2 = rnd(2 * B, 1, 7, 7, seed=1), rnd(2 * B, 1, 3, 3, seed=2)
    z1, z2 = rnd(2 * B, nc, seed=3), rnd(2 * B, nc, seed=4)
    lab = torch.randint(0, nc, (B,), generator=torch.Generator().manual_seed(5))
    wc = 0.7
    dev = [t.cuda().requires_grad_(True) for t in (o1, o2, z1, z2)]
    total, parts = ops.d_losses(dev[:2], dev[2:], lab.cuda(), B, 1.0, 0.0, wc)
    (total * 1.5).backward()
    ref = [t.clone().requires_grad_(True) for t in (o1, o2, z1, z2)]
    mse = torch.nn.MSELoss()
    onehot = torch.eye(nc)[lab]
    real = 0.5 * (mse(ref[0][:B], torch.ones_like(ref[0][:B])) + mse(ref[1][:B], torch.ones_like(ref[1][:B])))
    fake = 0.5 * (mse(ref[0][B:], torch.zeros_like(ref[0][B:])) + mse(ref[1][B:], torch.zeros_like(ref[1][B:])))
    cls = 0.5 * (mse(torch.softmax(ref[2][:B], 1), onehot) + mse(torch.softmax(ref[3][:B], 1), onehot))
    want = real + cls * wc + fake
    (want * 1.5).backward()
    close(total, want, 1e-5)
    close(parts, torch.stack([real, cls, fake]).detach(), 1e-5)
    for a, b in zip(dev, ref):
        close(a.grad, b.grad, 1e-5)
    # all rows "first", no class head (the generator's view of D without labels)
    t2, p2 = ops.d_losses([o1.cuda()], [], None, 2 * B, 1.0, 0.0, 0.0)
    close(t2, mse(o1, torch.ones_like(o1)), 1e-5)
    # lincomb
    xs = [rnd(1, seed=10 + i).reshape(()).cuda().requires_grad_(i != 1) for i in range(5)]
    ws = [1.0, 5.0, -0.25, 100.0, 0.5]
    out = ops.lincomb(list(zip(xs, ws)))
    (out * 2.0).backward()
    close(out, torch.tensor(sum(w * float(x) for x, w in zip(xs, ws))), 1e-6)
    for i, (x, w) in enumerate(zip(xs, ws)):
        if i == 1:
            assert x.grad is None
        else:
            close(x.grad, torch.tensor(2.0 * w), 1e-6)
    # conventional KL
    mu, lv = rnd(B, 8, seed=20), rnd(B, 8, seed=21) * 0.3
    a, b = mu.cuda().requires_grad_(True), lv.cuda().requires_grad_(True)
    kl = ops.kl_normal(a, b)
    kl.backward()
    ra, rb = mu.clone().requires_grad_(True), lv.clone().requires_grad_(True)
    want = -0.5 * torch.sum(1 + rb - ra ** 2 - rb.exp())
    want.backward()
    close(kl, want, 1e-5)
    close(a.grad, ra.grad, 1e-5)
    close(b.grad, rb.grad, 1e-5)


@pytest.mark.parametrize("o,i,k,s,p", [(128, 64, 4, 2, 1), (96, 40, 3, 1, 1), (512, 256, 4, 2, 1), (64, 3, 4, 2, 1)])
def test_multi_pack_launch_equals_single_pack_implicit_gemm(ops, o, i, k, s, p):
    """The implicit-GEMM operands ([phase][Npad][(tap, channel)] with zero padding) through the per-layer pack kernel and through
    the multi-layer launch: identical bytes, forward and input-gradient (per-output-phase) operands; the map is small enough
    that no Winograd form applies.  Both take the row-cooperative path (a (phase, n) row transposed through LDS) where it fits."""
    w = (rnd(o, i, k, k, seed=11) / np.sqrt(i * k * k)).cuda().requires_grad_(True)
    x = rnd(2, i, 8, 8, seed=12).cuda().requires_grad_(True)
    ops.invalidate_packed()
    try:
        with ops.pack_cache():
            y = ops.conv2d(x, w, None, s, p)
            y.sum().backward()
            hits = [h for h in ops._pack_cache.values() if h.weight is w]
            assert len(hits) == 2
            single = [h.buf.clone() for h in hits]
            for h in hits:
                h.buf.fill_(255)
            ops.refresh_packed([w], force=True)
            for h, sgl in zip(hits, single):
                if not h.fresh:          # an operand outside the multi-pack launch (3-channel layers): re-packed at its next use
                    assert ops._packed(h.desc, w, h.kind, h.act)[0] is h and h.fresh
                assert torch.equal(h.buf, sgl), (h.kind, o, i, k)
        ref = F.conv2d(x.detach().cpu(), w.detach().cpu(), None, s, p)
        close(y, ref, 2e-5)
    finally:
        ops.invalidate_packed()


@pytest.mark.parametrize("n,c0,c1,c2,hw,packed", [(8, 64, 128, 256, 64, True), (2, 8, 16, 24, 16, True), (2, 32, 64, 64, 32, False)])
def test_leaky_relu_backward_in_the_consumers_input_gradient(ops, n, c0, c1, c2, hw, packed):
    """conv(4x4, s2) + LeakyReLU(0.2) -> conv(4x4, s2): the first layer is told that its activation's backward is done by its
    consumer (``act_bwd_by_consumer``), the second one multiplies its input gradient by LeakyReLU'(its input)
    (``in_slope`` -> srgan_conv2d_dgrad_packed_mask: in the epilogue of the transposed F(3x3,2x2) kernel for the first shape,
    one in-place pass behind the implicit GEMM for the second, the unpacked entry points for the third).  Outputs and all three
    gradients against PyTorch on the CPU, and against the unchained call of the same kernels."""
    import contextlib
    x = rnd(n, c0, hw, hw, seed=31)
    w1 = rnd(c1, c0, 4, 4, seed=32) / np.sqrt(c0 * 16)
    w2 = rnd(c2, c1, 4, 4, seed=33) / np.sqrt(c1 * 16)
    gy = rnd(n, c2, hw // 4, hw // 4, seed=34)
    xr, w1r, w2r = (t.clone().requires_grad_(True) for t in (x, w1, w2))
    F.conv2d(F.leaky_relu(F.conv2d(xr, w1r, None, 2, 1), 0.2), w2r, None, 2, 1).backward(gy)
    res = {}
    ops.invalidate_packed()
    try:
        for chained in (False, True):
            xg, w1g, w2g = (t.clone().cuda().requires_grad_(True) for t in (x, w1, w2))
            with (ops.pack_cache() if packed else contextlib.nullcontext()):
                h = ops.conv2d(xg, w1g, None, 2, 1, ops.PAD_ZERO, ops.ACT_LRELU, 0.2, None, chained)
                y = ops.conv2d(h, w2g, None, 2, 1, ops.PAD_ZERO, ops.ACT_NONE, 0.0, 0.2 if chained else None, False)
                y.backward(gy.cuda())
            res[chained] = (y.detach(), xg.grad, w1g.grad, w2g.grad)
    finally:
        ops.invalidate_packed()
    for got, want in zip(res[True][1:], (xr.grad, w1r.grad, w2r.grad)):
        close(got, want, 3e-5)
    for a, b in zip(res[True], res[False]):
        close(a, b, 1e-6)


@pytest.mark.parametrize("n,c0,c1,c2,hw", [(2, 8, 64, 128, 128), (2, 16, 128, 256, 64), (2, 16, 128, 256, 32), (4, 8, 256, 512, 32)])
def test_leaky_relu_backward_in_the_consumers_input_gradient_bf16_mode(ops, n, c0, c1, c2, hw):
    """The same chain in the bf16 mode: the consumer's input gradient runs on halo16t_kernel (first shape) or igemm16_kernel
    (with and without split-K) followed by the in-place LeakyReLU-backward pass (fusing the mask into those epilogues was
    measured in round 4: the pass disappears, the kernels take as much longer -- profiles/LOG.md).  Chained and unchained calls
    multiply the same fp32 input gradient by the same 1 / 0.2, so they agree like the fp32 ones; both are held to the exact-fp32
    chain at bf16 accuracy."""
    x = rnd(n, c0, hw, hw, seed=31)
    w1 = rnd(c1, c0, 4, 4, seed=32) / np.sqrt(c0 * 16)
    w2 = rnd(c2, c1, 4, 4, seed=33) / np.sqrt(c1 * 16)
    gy = rnd(n, c2, hw // 4, hw // 4, seed=34)
    xr, w1r, w2r = (t.clone().requires_grad_(True) for t in (x, w1, w2))
    F.conv2d(F.leaky_relu(F.conv2d(xr, w1r, None, 2, 1), 0.2), w2r, None, 2, 1).backward(gy)
    res = {}
    ops.invalidate_packed()
    ops.set_compute_dtype("bf16")
    try:
        for chained in (False, True):
            xg, w1g, w2g = (t.clone().cuda().requires_grad_(True) for t in (x, w1, w2))
            with ops.pack_cache():
                h = ops.conv2d(xg, w1g, None, 2, 1, ops.PAD_ZERO, ops.ACT_LRELU, 0.2, None, chained)
                y = ops.conv2d(h, w2g, None, 2, 1, ops.PAD_ZERO, ops.ACT_NONE, 0.0, 0.2 if chained else None, False)
                y.backward(gy.cuda())
            res[chained] = (y.detach(), xg.grad, w1g.grad, w2g.grad)
    finally:
        ops.set_compute_dtype("fp32")
        ops.invalidate_packed()
    for a, b in zip(res[True], res[False]):
        close(a, b, 1e-6)
    for got, want in zip(res[True][1:], (xr.grad, w1r.grad, w2r.grad)):
        assert float((got.cpu() - want).norm() / want.norm()) < 2e-2          # bf16 operands (2^-9 each), a few sign flips of h


@pytest.mark.parametrize("hw", [(16, 128), (16, 64)])
def test_generator_bf16_activation_storage_vs_fp32_tensors(ops, hw):
    """(16 x 64, ADVICE r4: the trunk map is 4 x 16, so up conv 0 is NOT served by the 16-bit patch kernels while up conv 1 is --
    norm 0 of the up path then has to decide on its own shape whether it may write bf16.)
    Round 4: in the bf16 mode the tensors between the generator's stride-2 convolutions and their norms live in HBM as bf16
    (ops.py "bf16 activation storage": halo16s / halo16t / halo16s2_wgrad kernels with 16-bit I/O, instance norm with 16-bit
    I/O).  A full-width generator on a 16 x 128 image (so that every down / up layer is served: 8 x 64 and 4 x 32 maps) runs
    forward + backward with the storage on and off; both are bf16-mode results, so they are held to each other and, for scale,
    to the fp32-tensor chain's own distance from the exact-fp32 mode."""
    from srgan_amd import model
    torch.manual_seed(0)
    G = model.SingleGenerator(3, 64, 2, 2, 1, "instance", num_con=12).cuda()
    x = (torch.rand(2, 3, *hw) * 2 - 1).cuda()
    c = torch.cat([torch.eye(4)[torch.tensor([1, 3])], torch.randn(2, 8)], 1).cuda()
    gy = rnd(2, 3, *hw, seed=5).cuda()
    params = [p for p in G.parameters()]

    def run(mode, storage):
        ops.set_compute_dtype(mode)
        ops.STORAGE_BF16 = storage
        try:
            for p in params:
                p.grad = None
            with ops.pack_cache():
                if mode == "bf16":
                    probe = torch.empty((2, 64, *hw), device="meta")
                    assert G.down_convs[1].s2_io_applicable(probe) == storage      # the path under test is really taken
                    if hw == (16, 64) and storage:
                        t0, t1 = torch.empty((2, 256, 4, 16), device="meta"), torch.empty((2, 128, 8, 32), device="meta")
                        assert not G.up_convs[0].s2_io_applicable(t0) and G.up_convs[1].s2_io_applicable(t1)
                y = G(x, c)
                (y * gy).sum().backward()
            return y.detach().clone(), [p.grad.detach().clone() for p in params]
        finally:
            ops.STORAGE_BF16 = True
            ops.set_compute_dtype("fp32")
            ops.invalidate_packed()

    y32, g32 = run("fp32", True)
    yc, gc = run("bf16", False)          # bf16 products, fp32 tensors
    ys, gs = run("bf16", True)           # bf16 products, bf16 tensors between conv and norm

    # The yardstick (round 5, scratch/bf16_grad_errors.py): the exact-fp32 mode with the input and the weights rounded to bf16
    # ONCE -- a 2^-9 perturbation of the first operands, fp32 arithmetic everywhere.  It already moves the gradients of the early
    # layers by 11-13.5 % in relative L2 (output: 0.55 %): 0.06-0.3 % of the ReLU masks of every norm + ReLU flip (measured per
    # layer), a flipped element switches its whole gradient path on or off, and under ~12 such layers every upstream gradient is a
    # sum over flipped and unflipped paths -- so the error is broad (median error / median magnitude 0.16), unlike the single
    # flips close_grad handles.  The bf16 mode rounds the operands of EVERY convolution and sits at 1.0-1.6x that: the 0.169 that
    # broke the former 0.15 bound is down_cnorms.0.ConBias.0.weight (0.165 here; 0.141 with the RGB input layer on the fp32
    # kernel, SRGAN_NO_RGBIN16 in the exp build: its bf16 products perturb the first pre-activations) -- the network's
    # sensitivity to operand rounding, not an error of the 16-bit kernels, which are held to the fp32 convolution of the
    # bf16-rounded operands at 2e-5 one by one (test_rgb_input_form_bf16_compute_mode, test_conv2d_bf16_compute_mode).
    def bf16r(t):
        return t.to(torch.bfloat16).to(torch.float32)
    keep, x_keep = [p.data.clone() for p in params], x.clone()
    try:
        for p in params:
            p.data.copy_(bf16r(p.data))
        x.copy_(bf16r(x))
        yp, gp = run("fp32", True)
    finally:
        for p, k in zip(params, keep):
            p.data.copy_(k)
        x.copy_(x_keep)

    def rel(a, b):
        return float((a - b).norm() / (b.norm() + 1e-12))

    assert rel(ys, y32) <= 1.5 * rel(yc, y32) + 2e-3, (rel(ys, y32), rel(yc, y32))
    assert rel(ys, y32) <= 2e-2
    worst = ("", 0.0)
    for (name, _), a, b, q, r in zip(G.named_parameters(), gs, gc, gp, g32):
        e_s, e_c, e_p = rel(a, r), rel(b, r), rel(q, r)
        if e_s > worst[1]:
            worst = (name, e_s)
        # storage on against storage off: the same mask-flip noise on both sides
        assert e_s <= 1.6 * e_c + 1e-2, (name, e_s, e_c)
        # against the yardstick: measured e_s / e_p <= 1.58 over all 46 tensors (up_convs.2.weight; 1.47 on the 15 % tensors)
        assert e_s <= 2.0 * e_p + 5e-3, (name, e_s, e_p)
    if hw == (16, 128):
        assert worst[1] <= 0.215, worst      # 1.3 x the largest measured value (0.1653, down_cnorms.0.ConBias.0.weight)


def test_sink_slot_that_aliases_a_live_gradient_accumulates(ops):
    """under data parallelism the gradient sink hands out a parameter's persistent BUCKET SLICE
    (`ops._sink_alloc` = `dp.grad_slot`), which is the very tensor an earlier pass bound to `p.grad`.  Two backward passes
    without `zero_grad` between them (gradient accumulation) must ADD: the first weight-gradient launch of the second pass used
    to run with accumulate = False over the live gradient, and the scope's exit -- `p.grad is buf` -- skipped the add.  A stand-in
    allocator plays `dp.grad_slot` here (no process group needed): conv weight + bias, one and two passes."""
    torch.manual_seed(3)
    w = (torch.randn(64, 32, 3, 3) / 17).cuda().requires_grad_(True)
    b = torch.randn(64).cuda().requires_grad_(True)
    x = torch.randn(4, 32, 20, 20).cuda().contiguous(memory_format=torch.channels_last)
    gy = torch.randn(4, 64, 20, 20).cuda().contiguous(memory_format=torch.channels_last)
    slots = {}

    def alloc(p):
        return slots.setdefault(id(p), torch.zeros_like(p))

    def one_pass():
        y = ops.conv2d(x, w, b, 1, 1)
        with ops.fused_param_grads():
            y.backward(gy)

    one_pass()
    g1w, g1b = w.grad.clone(), b.grad.clone()
    w.grad = b.grad = None
    prev = ops._sink_alloc
    ops._sink_alloc = alloc
    try:
        one_pass()
        assert w.grad.data_ptr() == slots[id(w)].data_ptr()           # the slot IS the gradient now
        close(w.grad, g1w)
        one_pass()                                                     # no zero_grad: the slot is live
        close(w.grad, 2 * g1w)
        close(b.grad, 2 * g1b)
        assert w.grad.data_ptr() == slots[id(w)].data_ptr()
        w.grad = b.grad = None                                         # after zero_grad the slot is overwritten again
        one_pass()
        close(w.grad, g1w)
        close(b.grad, g1b)
    finally:
        ops._sink_alloc = prev


def test_encoder_bf16_activation_storage_vs_fp32_tensors(ops):
    """Round 5: in the bf16 mode the tensors inside the style encoder's blocks (norm -> conv -> norm -> conv -> pool) live in
    HBM as bf16 (ops.py "16-bit activations around the generic convolutions").  The full-width encoder on 64 x 64 images (30 /
    15 / 7 / 3-pixel maps: the first three blocks are served, the 3 x 3 map's reflect convolutions too) runs forward + backward
    with the storage on and off; both are bf16-mode results, so they are held to each other and, for scale, to the fp32-tensor
    chain's own distance from the exact-fp32 mode and to the exact-fp32 mode with the operands rounded to bf16 once."""
    from srgan_amd import model
    torch.manual_seed(0)
    E = model.Encoder(3, 8, 64, 4, "instance", 4, "cuda").cuda()
    x = (torch.rand(4, 3, 64, 64) * 2 - 1).cuda()
    gmu, glv, gcl = rnd(4, 8, seed=6).cuda(), rnd(4, 8, seed=7).cuda(), rnd(4, 4, seed=8).cuda()
    params = [p for p in E.parameters()]

    def run(mode, storage):
        ops.set_compute_dtype(mode)
        ops.STORAGE_BF16 = storage
        try:
            for p in params:
                p.grad = None
            with ops.pack_cache():
                if mode == "bf16":
                    probe = torch.empty((4, 64, 30, 30), device="cuda")
                    blk = E.layers[0]
                    assert model._block_io16(probe, blk.conv1, blk.cmp[0]) == storage      # the path under test is really taken
                _, mu, logvar, cls, _ = E(x)
                ((mu * gmu).sum() + (logvar * glv).sum() + (cls * gcl).sum()).backward()
            return torch.cat([mu, logvar, cls], 1).detach().clone(), [p.grad.detach().clone() for p in params]
        finally:
            ops.STORAGE_BF16 = True
            ops.set_compute_dtype("fp32")
            ops.invalidate_packed()

    y32, g32 = run("fp32", True)
    yc, gc = run("bf16", False)          # bf16 products, fp32 tensors
    ys, gs = run("bf16", True)           # bf16 products, bf16 tensors inside the blocks

    def bf16r(t):
        return t.to(torch.bfloat16).to(torch.float32)
    keep, x_keep = [p.data.clone() for p in params], x.clone()
    try:
        for p in params:
            p.data.copy_(bf16r(p.data))
        x.copy_(bf16r(x))
        yp, gp = run("fp32", True)          # the yardstick of the generator's twin below
    finally:
        for p, k in zip(params, keep):
            p.data.copy_(k)
        x.copy_(x_keep)

    def rel(a, b):
        return float((a - b).norm() / (b.norm() + 1e-12))

    print("encoder output: storage %.3e, fp32 tensors %.3e, rounded-once %.3e" % (rel(ys, y32), rel(yc, y32), rel(yp, y32)))
    assert rel(ys, y32) <= 1.5 * rel(yc, y32) + 2e-3, (rel(ys, y32), rel(yc, y32))
    assert rel(ys, y32) <= 2e-2
    worst = ("", 0.0, 0.0, 0.0)
    for (name, _), a, b, q, r in zip(E.named_parameters(), gs, gc, gp, g32):
        e_s, e_c, e_p = rel(a, r), rel(b, r), rel(q, r)
        if e_s > worst[1]:
            worst = (name, e_s, e_c, e_p)
        assert e_s <= 1.6 * e_c + 1e-2, (name, e_s, e_c)
        assert e_s <= 2.0 * e_p + 5e-3, (name, e_s, e_p)
    print("encoder gradients, worst tensor:", worst)
    # an absolute cap beside the relative ones, as the generator's twin has -- 1.3 x the largest measured
    # value (0.28 on this shape; its rounded-once yardstick is 0.26)
    assert worst[1] <= 0.365, worst


@pytest.mark.parametrize("in16", [False, True])
@pytest.mark.parametrize("out16", [False, True])
def test_stride2_io_functions_every_dtype_pair(ops, in16, out16):
    """conv2d_s2_io / conv_transpose2d_io are documented for any fp32 / bf16 mix of input and output; the weight
    gradient of the (fp32 input side, bf16 output side) pair was not instantiated, so two of the four combinations passed
    forward and failed in backward.  All four, both functions, forward + input gradient + weight gradient, against the fp32
    convolution of the bf16-rounded operands (a bf16 result is compared after the same rounding of the reference)."""
    torch.set_num_threads(16)
    n, ci, co, h, w = 2, 64, 128, 16, 64
    x = rnd(n, ci, h, w, seed=21)
    wt = rnd(co, ci, 4, 4, seed=22) / np.sqrt(ci * 16)
    xt = rnd(n, co, h // 2, w // 2, seed=23)                   # input of the transposed conv (weight viewed [Cin = co][Cout = ci])
    gy = rnd(n, co, h // 2, w // 2, seed=24)
    gyt = rnd(n, ci, h, w, seed=25)

    def r16(t, on):
        return _bf16_round(t) if on else t

    def run_ref(fn, inp, g):
        iv, wv = inp.clone().requires_grad_(True), _bf16_round(wt).clone().requires_grad_(True)
        out = fn(iv, wv)
        out.backward(g)
        return out.detach(), iv.grad, wv.grad

    ops.set_compute_dtype("bf16")
    try:
        with ops.pack_cache():
            wd = wt.cuda().requires_grad_(True)
            probe = torch.empty((n, ci, h, w), device="meta")
            if not ops.s2_io_applicable(n, ci, h, w, co, wd, False):
                pytest.skip("patch kernels do not serve this shape")
            for fn_dev, fn_ref, inp, g in ((ops.conv2d_s2_io, lambda a, b: F.conv2d(a, b, None, 2, 1), x, gy),
                                           (ops.conv_transpose2d_io, lambda a, b: F.conv_transpose2d(a, b, None, 2, 1), xt, gyt)):
                # operands as the kernels see them: bf16-rounded input, weight and upstream gradient
                y_ref, dx_ref, dw_ref = run_ref(fn_ref, _bf16_round(inp), _bf16_round(g))
                wd.grad = None
                xd = inp.cuda().contiguous(memory_format=torch.channels_last)
                xd = (xd.to(torch.bfloat16) if in16 else xd).requires_grad_(True)
                y = fn_dev(xd, wd, out16)
                assert y.dtype == (torch.bfloat16 if out16 else torch.float32)
                y.backward(g.cuda().contiguous(memory_format=torch.channels_last).to(y.dtype))
                close(y.float(), r16(y_ref, out16), 2e-5 if not out16 else 5e-3)
                close(xd.grad.float(), r16(dx_ref, in16), 2e-5 if not in16 else 5e-3)
                close(wd.grad, dw_ref, 5e-5)
    finally:
        ops.set_compute_dtype("fp32")
        ops.invalidate_packed()


@pytest.mark.parametrize("in16", [False, True])
@pytest.mark.parametrize("out16", [False, True])
@pytest.mark.parametrize("shape", [(2, 64, 128, 64, 64), (3, 128, 256, 32, 32), (4, 256, 512, 16, 16), (2, 64, 128, 16, 16),
                                   (2, 512, 512, 16, 16)])
def test_conv_act_io_every_dtype_pair(ops, shape, in16, out16):
    """Round 6 (VERDICT r5 item 1 iii): the discriminator trunks' conv 4x4 / stride 2 / pad 1 (no bias) + LeakyReLU(0.01) layers
    (pyfiles/model.py:302-309; d1: 64 -> 128 @ 64, 128 -> 256 @ 32, 256 -> 512 @ 16, at 256 x 256 also 512 -> 512; d2: 64 -> 128
    @ 16) with bf16 tensors on either side in the bf16 mode (ops.conv2d_act_io: halo16s_kernel / igemm16_kernel with LDS-DMA tiles
    forward, halo16t_kernel / igemm16_kernel backward, halo16s2_wgrad_kernel): forward + input gradient + weight gradient of
    every fp32 / bf16 mix against the fp32 convolution of the bf16-rounded operands.  The LeakyReLU derivative is applied to the
    incoming gradient and the product rounded to bf16 (what the device's act_bwd_io writes) before the reference's backward."""
    torch.set_num_threads(16)
    n, ci, co, h, w = shape
    slope = 0.01
    x = rnd(n, ci, h, w, seed=41)
    wt = rnd(co, ci, 4, 4, seed=42) / np.sqrt(ci * 16)
    gy = rnd(n, co, h // 2, w // 2, seed=43)

    def r16(t, on):
        return _bf16_round(t) if on else t

    ops.set_compute_dtype("bf16")
    try:
        with ops.pack_cache():
            wd = wt.cuda().requires_grad_(True)
            if not ops.conv_act_io_applicable(n, ci, h, w, wd, ops.ACT_LRELU):
                pytest.skip("no 16-bit path for this shape")
            xd = x.cuda().contiguous(memory_format=torch.channels_last)
            xd = (xd.to(torch.bfloat16) if in16 else xd).requires_grad_(True)
            y = ops.conv2d_act_io(xd, wd, ops.ACT_LRELU, slope, out16)
            assert y.dtype == (torch.bfloat16 if out16 else torch.float32)
            g_dev = gy.cuda().contiguous(memory_format=torch.channels_last).to(y.dtype)
            y.backward(g_dev)
            # reference: operands as the kernels see them
            xr, wr = _bf16_round(x).clone().requires_grad_(True), _bf16_round(wt).clone().requires_grad_(True)
            z = F.conv2d(xr, wr, None, 2, 1)
            y_ref = F.leaky_relu(z, slope)
            close(y.float().cpu(), r16(y_ref.detach(), out16), 2e-5 if not out16 else 5e-3)
            # the device multiplies the gradient it received (of y's type) by the derivative taken from the y it STORED and rounds
            # the product to bf16
            y_seen = y.detach().float().cpu()
            g = _bf16_round(r16(gy, out16) * torch.where(y_seen > 0, torch.ones_like(y_seen), torch.full_like(y_seen, slope)))
            z.backward(g)
            close(xd.grad.float().cpu(), r16(xr.grad, in16), 2e-5 if not in16 else 5e-3)
            close(wd.grad.cpu(), wr.grad, 5e-5)
    finally:
        ops.set_compute_dtype("fp32")
        ops.invalidate_packed()


@pytest.mark.parametrize("wide16", [False, True])
@pytest.mark.parametrize("layer", ["in", "out"])
@pytest.mark.parametrize("n,h,w", [(2, 32, 64), (3, 80, 96)])
def test_rgb_layers_with_a_bf16_64_channel_side(ops, n, h, w, layer, wide16):
    """Round 6: the generator's 7x7 / pad-3 RGB layers (pyfiles/model.py:212, 232) through ops.conv2d_act_io with their 64-channel
    side in bf16 (the 3-channel side is always fp32): input layer = rgbin16_conv_kernel<OUT16> forward, rgbout16_conv_kernel<IN16>
    input gradient, rgb_wgrad16_kernel<0, C16>; output layer = rgbout16<IN16> forward, rgbin16<OUT16> input gradient,
    rgb_wgrad16_kernel<1, C16>.  Every product against the fp32 convolution of the bf16-rounded operands; the second shape has a
    ragged last column strip (96 = 3 x 32 but 80 rows = 5 blocks of 16) and 3 images."""
    torch.set_num_threads(16)
    ci, co = (3, 64) if layer == "in" else (64, 3)
    x = rnd(n, ci, h, w, seed=51)
    wt = rnd(co, ci, 7, 7, seed=52) / np.sqrt(ci * 49)
    gy = rnd(n, co, h, w, seed=53)
    in16, out16 = (False, wide16) if layer == "in" else (wide16, False)

    def r16(t, on):
        return _bf16_round(t) if on else t

    ops.set_compute_dtype("bf16")
    try:
        with ops.pack_cache():
            wd = wt.cuda().requires_grad_(True)
            assert ops.conv_act_io_applicable(n, ci, h, w, wd, ops.ACT_NONE, 7, 1, 3)
            xd = x.cuda().contiguous(memory_format=torch.channels_last)
            xd = (xd.to(torch.bfloat16) if in16 else xd).requires_grad_(True)
            y = ops.conv2d_act_io(xd, wd, ops.ACT_NONE, 0.0, out16, 7, 1, 3)
            assert y.dtype == (torch.bfloat16 if out16 else torch.float32)
            y.backward(gy.cuda().contiguous(memory_format=torch.channels_last).to(y.dtype))
            xr, wr = _bf16_round(x).clone().requires_grad_(True), _bf16_round(wt).clone().requires_grad_(True)
            z = F.conv2d(xr, wr, None, 1, 3)
            close(y.float().cpu(), r16(z.detach(), out16), 2e-5 if not out16 else 5e-3)
            z.backward(_bf16_round(gy))
            close(xd.grad.float().cpu(), r16(xr.grad, in16), 2e-5 if not in16 else 5e-3)
            close(wd.grad.cpu(), wr.grad, 5e-5)
            # the wrong side in bf16 is refused, not silently converted
            if layer == "in":
                with pytest.raises(Exception, match="fp32"):
                    ops.conv2d_act_io(x.cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last), wd, ops.ACT_NONE, 0.0, False, 7, 1, 3)
            else:
                with pytest.raises(Exception, match="fp32"):
                    ops.conv2d_act_io(x.cuda().contiguous(memory_format=torch.channels_last), wd, ops.ACT_NONE, 0.0, True, 7, 1, 3)
    finally:
        ops.set_compute_dtype("fp32")
        ops.invalidate_packed()


def test_discriminator_bf16_activation_storage_vs_fp32_tensors(ops):
    """Round 6: the full-width discriminator (both scales, heads) forward + backward in the bf16 mode with the trunks' 16-bit
    activations on and off: both are bf16-mode results (same products; the storage adds one rounding per stored activation and
    gradient), held to each other and to the fp32-tensor chain's own distance from the exact-fp32 mode."""
    from srgan_amd import model
    torch.manual_seed(0)
    D = model.SingleDiscriminator_solo_multi(3, 64, 2, 4, "instance", 4).cuda()
    x = (torch.rand(4, 3, 128, 128) * 2 - 1).cuda()
    params = [p for p in D.parameters()]

    def run(mode, storage):
        ops.set_compute_dtype(mode)
        ops.STORAGE_BF16 = storage
        try:
            for p in params:
                p.grad = None
            xg = x.clone().requires_grad_(True)
            with ops.pack_cache():
                if mode == "bf16":
                    probe = torch.empty((4, 64, 64, 64), device="cuda")
                    assert D.discriminator1.down_convs[2].act_io_applicable(probe, ops.ACT_LRELU) == storage
                outs, logits = D.forward_logits(xg)
                loss = sum((o ** 2).mean() for o in outs) + sum((z * torch.linspace(-1, 1, z.numel(), device=z.device).view_as(z)).sum()
                                                                for z in logits)
                loss.backward()
            return ([o.detach().clone() for o in outs] + [z.detach().clone() for z in logits], xg.grad.detach().clone(),
                    [p.grad.detach().clone() for p in params])
        finally:
            ops.STORAGE_BF16 = True
            ops.set_compute_dtype("fp32")
            ops.invalidate_packed()

    o32, dx32, g32 = run("fp32", True)
    oc, dxc, gc = run("bf16", False)
    os_, dxs, gs = run("bf16", True)

    def rel(a, b):
        return float((a - b).norm() / (b.norm() + 1e-12))

    for a, b, r in zip(os_, oc, o32):
        assert rel(a, r) <= 1.5 * rel(b, r) + 3e-3, (rel(a, r), rel(b, r))
        assert rel(a, r) <= 2e-2
    assert rel(dxs, dx32) <= 1.5 * rel(dxc, dx32) + 5e-3, (rel(dxs, dx32), rel(dxc, dx32))
    worst = ("", 0.0, 0.0)
    for (name, _), a, b, r in zip(D.named_parameters(), gs, gc, g32):
        e_s, e_c = rel(a, r), rel(b, r)
        if e_s > worst[1]:
            worst = (name, e_s, e_c)
        assert e_s <= 1.6 * e_c + 1e-2, (name, e_s, e_c)
    print("discriminator gradients, worst tensor:", worst, "input gradient:", rel(dxs, dx32), rel(dxc, dx32))
    # measured: 0.0789 with the storage on and 0.0789 with it off (discriminator1.down_convs.2.weight): the distance is the bf16
    # mode's, not the storage's; 1.3 x as the absolute cap
    assert worst[1] <= 0.103, worst


@pytest.mark.parametrize("in16", [False, True])
@pytest.mark.parametrize("out16", [False, True])
@pytest.mark.parametrize("shape", [(2, 64, 128, 30, 30, "reflect"), (3, 128, 128, 15, 15, "reflect"), (4, 256, 512, 7, 7, "reflect"),
                                   (2, 128, 64, 12, 20, "zeros"),
                                   # round 6: the same Functions with the encoder's large-map layers on halo16e_kernel (the
                                   # dispatch threshold switched off so that these small batches take it): every (Cin, Cout) it
                                   # is instantiated for, ragged patches in both directions, reflect (input gradient = padded
                                   # gradient + fold) and zero padding (input gradient straight into the tensor)
                                   (2, 64, 64, 30, 30, "reflect", "halo"), (2, 64, 128, 62, 62, "reflect", "halo"),
                                   (3, 128, 128, 31, 31, "reflect", "halo"), (2, 128, 256, 31, 31, "reflect", "halo"),
                                   (2, 64, 128, 20, 40, "zeros", "halo"), (1, 128, 128, 9, 33, "zeros", "halo"),
                                   (2, 128, 256, 8, 64, "zeros", "halo"), (1, 64, 64, 33, 5, "reflect", "halo")])
def test_generic_conv_io_every_dtype_pair(ops, shape, in16, out16):
    """Round 5: the 3x3 stride-1 layers of the style encoder (reflect-padded, 62 / 31 / 15 / 7-pixel maps; model.py:413-437) with
    bf16 tensors on either side in the bf16 mode (igemm16_kernel<IN16, OUT16>, its split-K sum, the reflect fold, wgrad_kernel
    with 16-bit loads): forward + input gradient + weight gradient of every fp32 / bf16 mix against the fp32 convolution of the
    bf16-rounded operands (a bf16 result is compared after the same rounding of the reference).  The 7 x 7 case runs split-K."""
    torch.set_num_threads(16)
    n, ci, co, h, w, pm = shape[:6]
    halo = len(shape) > 6
    if halo:
        os.environ["SRGAN_WINOGRAD_THRESHOLD_SCALE"] = "0"
    x = rnd(n, ci, h, w, seed=31)
    wt = rnd(co, ci, 3, 3, seed=32) / np.sqrt(ci * 9)
    gy = rnd(n, co, h, w, seed=33)

    def r16(t, on):
        return _bf16_round(t) if on else t

    iv, wv = _bf16_round(x).clone().requires_grad_(True), _bf16_round(wt).clone().requires_grad_(True)
    xp = F.pad(iv, (1, 1, 1, 1), mode="reflect") if pm == "reflect" else F.pad(iv, (1, 1, 1, 1))
    y_ref = F.conv2d(xp, wv)
    y_ref.backward(_bf16_round(gy))
    ops.set_compute_dtype("bf16")
    try:
        with ops.pack_cache():
            wd = wt.cuda().requires_grad_(True)
            mode = ops.PAD_REFLECT if pm == "reflect" else ops.PAD_ZERO
            if not ops.conv_io_applicable(n, ci, h, w, wd, 1, mode):
                pytest.skip("generic bf16 kernels do not serve this shape")
            xd = x.cuda().contiguous(memory_format=torch.channels_last)
            xd = (xd.to(torch.bfloat16) if in16 else xd).requires_grad_(True)
            y = ops.conv2d_io(xd, wd, 1, mode, out16)
            assert y.dtype == (torch.bfloat16 if out16 else torch.float32)
            y.backward(gy.cuda().contiguous(memory_format=torch.channels_last).to(y.dtype))
            assert xd.grad.dtype == xd.dtype
            close(y.float(), r16(y_ref.detach(), out16), 2e-5 if not out16 else 5e-3)
            close(xd.grad.float(), r16(iv.grad, in16), 2e-5 if not in16 else 5e-3)
            close(wd.grad, wv.grad, 5e-5)
            # the same layer through the fp32-tensor entry points of the bf16 mode: same kernels, same sums
            x2 = _bf16_round(x).cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
            w2 = wt.cuda().requires_grad_(True)
            y2 = ops.conv2d(x2, w2, None, 1, 1, mode)
            y2.backward(_bf16_round(gy).cuda().contiguous(memory_format=torch.channels_last))
            if out16:
                assert torch.equal(y, y2.to(torch.bfloat16))
            else:
                assert torch.equal(y, y2)
            assert torch.equal(wd.grad, w2.grad)
    finally:
        os.environ.pop("SRGAN_WINOGRAD_THRESHOLD_SCALE", None)
        ops.set_compute_dtype("fp32")
        ops.invalidate_packed()


@pytest.mark.parametrize("in16", [False, True])
@pytest.mark.parametrize("out16", [False, True])
def test_avgpool2_io_every_dtype_pair(ops, in16, out16):
    """AvgPool2d(2, 2) with fp32 / bf16 tensors on either side (odd map: the last row / column is dropped, its gradient zero)."""
    x = rnd(3, 64, 15, 31, seed=41)
    g = rnd(3, 64, 7, 15, seed=42)
    xr = (_bf16_round(x) if in16 else x).clone().requires_grad_(True)
    y_ref = F.avg_pool2d(xr, 2, 2)
    y_ref.backward(_bf16_round(g) if out16 else g)
    xd = x.cuda().contiguous(memory_format=torch.channels_last)
    xd = (xd.to(torch.bfloat16) if in16 else xd).requires_grad_(True)
    y = ops.avgpool2_io(xd, out16)
    assert y.dtype == (torch.bfloat16 if out16 else torch.float32)
    y.backward(g.cuda().contiguous(memory_format=torch.channels_last).to(y.dtype))
    assert xd.grad.dtype == xd.dtype
    ref_y = _bf16_round(y_ref.detach()) if out16 else y_ref.detach()
    ref_g = _bf16_round(xr.grad) if in16 else xr.grad
    assert torch.equal(y.float().cpu(), ref_y)
    assert torch.equal(xd.grad.float().cpu(), ref_g)


@pytest.mark.parametrize("n,hh,ww", [(2, 32, 64), (5, 64, 416)])
def test_rgb_input_form_bf16_compute_mode(ops, n, hh, ww):
    """Round 4: in the bf16 mode the 3 -> 64 channel 7x7 form -- the generator's RGB input layer forward and the input gradient
    of its RGB output layer -- runs on rgbin16_conv_kernel (bf16 MFMA, halo with 4 channels per pixel) and the 64 -> 3 head's
    forward on rgbout16_conv_kernel: equal to the fp32 convolution of the bf16-ROUNDED operands.  Round 6: so do the weight
    gradients of both layers (rgb_wgrad16_kernel: dy * x with both rounded to bf16, fp32 accumulation; they were the last
    exact-fp32 products of these layers in this mode).  The second shape has 1040 patches of 4 x 32 pixels in 347 ranges of
    3 (the last one 2): the double-buffered patch loop and a ragged last range."""
    torch.set_num_threads(16)
    x = rnd(n, 3, hh, ww, seed=11)
    w_in = rnd(64, 3, 7, 7, seed=12) / np.sqrt(147)
    h = rnd(n, 64, hh, ww, seed=13)
    w_out = rnd(3, 64, 7, 7, seed=14) / np.sqrt(3136)
    gy_in, gy_out = rnd(n, 64, hh, ww, seed=15), rnd(n, 3, hh, ww, seed=16)
    y_ref = F.conv2d(_bf16_round(x), _bf16_round(w_in), None, 1, 3)
    hr = h.clone().requires_grad_(True)
    F.conv2d(hr, _bf16_round(w_out), None, 1, 3).backward(_bf16_round(gy_out))
    # weight gradients: dW = wgrad(bf16(x), bf16(dy))
    wr = w_in.clone().requires_grad_(True)
    F.conv2d(_bf16_round(x), wr, None, 1, 3).backward(_bf16_round(gy_in))
    wor = w_out.clone().requires_grad_(True)
    F.conv2d(_bf16_round(h), wor, None, 1, 3).backward(_bf16_round(gy_out))
    ops.set_compute_dtype("bf16")
    try:
        for cached in (False, True):
            xd, wd = x.cuda(), w_in.cuda().requires_grad_(True)
            hd, wo = h.cuda().requires_grad_(True), w_out.cuda().requires_grad_(True)
            ctx = ops.pack_cache() if cached else contextlib.nullcontext()
            with ctx:
                y = ops.conv2d(xd, wd, None, 1, 3)
                y.backward(gy_in.cuda())
                yo = ops.conv2d(hd, wo, None, 1, 3)
                yo.backward(gy_out.cuda())
            close(y, y_ref)
            close(hd.grad, hr.grad)
            close(wd.grad, wr.grad)
            close(yo, F.conv2d(_bf16_round(h), _bf16_round(w_out), None, 1, 3))      # rgbout16_conv_kernel
            close(wo.grad, wor.grad)
    finally:
        ops.set_compute_dtype("fp32")
        ops.invalidate_packed()


def test_instance_norm_in_place_call_takes_the_separate_finalize(ops):
    """with the statistics finalize folded into the apply pass every workgroup re-reads the shift sample x[n][0][c]
    while workgroup 0 may already be storing y there -- an in-place call (y == x) through the C ABI now takes the separate
    finalize launch and must give the out-of-place result (to fp32 rounding)."""
    import ctypes
    from srgan_amd import _lib
    lib = _lib.load()
    n, c, h, w = 4, 64, 96, 96                     # two-pass shape (9216 pixels), C | 1024
    x = (rnd(n, h, w, c, seed=21) * 3 + 1).cuda()
    sc, sh = (1 + 0.2 * rnd(n, c, seed=22)).cuda(), (0.3 * rnd(n, c, seed=23)).cuda()
    nb = lib.srgan_instnorm_workspace(n, h * w, c)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())

    def run(src, dst):
        mean, rstd = torch.empty(n * c, device="cuda"), torch.empty(n * c, device="cuda")
        _lib.check(lib.srgan_instnorm_fwd(P(src), P(sc), P(sh), None, P(dst), P(mean), P(rstd), n, h * w, c, 1e-5, ops.ACT_RELU, 0.0,
                                          P(ws), nb, st), "instnorm_fwd")
        return mean, rstd
    y = torch.empty_like(x)
    m0, r0 = run(x, y)
    xi = x.clone()
    m1, r1 = run(xi, xi)
    torch.cuda.synchronize()
    # (the stand-alone finalize kernel may contract the same expression differently: equal to fp32 rounding, not bit for bit)
    close(m1, m0, 1e-6)
    close(r1, r0, 1e-6)
    close(xi, y, 2e-6)


def test_wgrad_arena_is_one_per_device(ops):
    """the deferred slab sums' arena was keyed by (device, raw stream pointer) -- every re-recorded step (a new capture
    stream each time) took another GiB.  One arena per device, whatever stream opens the scope."""
    arena_keys = lambda: [k for k in ops._workspaces if k[0] == "wgrad_arena"]
    saved = ops._WGRAD_ARENA_BYTES
    for k in arena_keys():
        del ops._workspaces[k]
    ops._WGRAD_ARENA_BYTES = 8 << 20
    try:
        w = (rnd(64, 64, 3, 3, seed=1) / 24).cuda().requires_grad_(True)
        x = rnd(2, 64, 16, 16, seed=2).cuda()
        for stream in (torch.cuda.current_stream(), torch.cuda.Stream(), torch.cuda.Stream()):
            stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(stream):
                w.grad = None
                with ops.pack_cache(), ops.fused_param_grads(True):
                    ops.conv2d(x, w, None, 1, 1).sum().backward()
            torch.cuda.current_stream().wait_stream(stream)
        torch.cuda.synchronize()
        assert len(arena_keys()) == 1, arena_keys()
        # sized from use: a pass that asks for more than the arena holds still gives the right gradient (early
        # flush) and makes the next scope take an arena that fits; after that it does not grow again
        ops._WGRAD_ARENA_BYTES = 1 << 20
        ops._arena_want.clear()
        for k in arena_keys():
            del ops._workspaces[k]
        wb = (rnd(256, 256, 3, 3, seed=3) / 48).cuda().requires_grad_(True)
        xb = rnd(8, 256, 32, 32, seed=4).cuda()
        sizes, grads = [], []
        for _ in range(3):
            wb.grad = None
            with ops.pack_cache(), ops.fused_param_grads(True):
                ops.conv2d(xb, wb, None, 1, 1).sum().backward()
            sizes.append(ops._workspaces[arena_keys()[0]].numel())
            grads.append(wb.grad.clone())
        assert sizes[1] > sizes[0] and sizes[2] == sizes[1], sizes
        assert sizes[1] >= ops._arena_want[torch.cuda.current_device()]
        assert torch.equal(grads[0], grads[1]) and torch.equal(grads[1], grads[2])
    finally:
        ops._WGRAD_ARENA_BYTES = saved
        ops._arena_want.clear()
        for k in arena_keys():
            del ops._workspaces[k]
        ops.invalidate_packed()


@pytest.mark.parametrize("shape", [(3, 64, 96, 96), (32, 128, 16, 16)])      # two-pass kernels; slab kernels (32 x 4 slabs)
@pytest.mark.parametrize("io", [(False, True), (True, True), (True, False)])
def test_instance_norm_16bit_io(ops, shape, io):
    """Round 4: instance norm with an fp32 or bf16 input and a bf16 or fp32 output (the norms between the generator's stride-2
    convolutions in the bf16 mode): equal to the fp32 op on the bf16-ROUNDED input, output rounded once; the backward takes the
    gradient in the output's type and returns dx in the input's."""
    n, c, h, w = shape
    x16, y16 = io
    ops.set_compute_dtype("bf16")
    try:
        assert ops.norm_io_applicable(n, c, h, w)
        x = rnd(n, c, h, w, seed=1) * 2 + 0.5
        sc, sh = 1 + 0.3 * rnd(n, c, seed=2), 0.5 * rnd(n, c, seed=3)
        gy = rnd(n, c, h, w, seed=4)
        xr = (_bf16_round(x) if x16 else x).clone().requires_grad_(True)
        scr, shr = sc.clone().requires_grad_(True), sh.clone().requires_grad_(True)
        yr = torch.relu(F.instance_norm(xr, eps=1e-5) * scr[:, :, None, None] + shr[:, :, None, None])
        gyr = _bf16_round(gy) if y16 else gy
        yr.backward(gyr)
        nhwc = lambda t, dt: t.permute(0, 2, 3, 1).contiguous().to(dt).cuda().permute(0, 3, 1, 2)
        xd = nhwc(x, torch.bfloat16 if x16 else torch.float32).requires_grad_(True)
        scd, shd = sc.cuda().requires_grad_(True), sh.cuda().requires_grad_(True)
        y = ops.instance_norm_act_io(xd, scd, shd, ops.ACT_RELU, 0.0, 1e-5, y16)
        assert y.dtype == (torch.bfloat16 if y16 else torch.float32)
        y.backward(nhwc(gy, y.dtype))
        assert xd.grad.dtype == xd.dtype
        tol_y = 6e-3 if y16 else 2e-5                       # one bf16 rounding of the result: 2^-8 relative
        close(y.float(), yr, tol_y)
        close(xd.grad.float(), xr.grad, 6e-3 if x16 else 1e-4)
        close(scd.grad, scr.grad, 1e-4)
        close(shd.grad, shr.grad, 1e-4)
    finally:
        ops.set_compute_dtype("fp32")

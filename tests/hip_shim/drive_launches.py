"""Runs under LD_PRELOAD=launch_shim.so (tests/test_launch_plan_cpu.py): calls the convolution / norm entry points of the C ABI
for every layer of G / D / E at one BASELINE geometry, in both compute modes, with fake device pointers.  Nothing executes; the
shim logs every launch descriptor the host code builds.  usage: drive_launches.py <lib> <H> <B> <dis_num_cls>"""
import ctypes
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "style-restricted_gan_amd"))
from srgan_amd import _lib                                                     # noqa: E402  (ctypes only, no torch)

FAKE = 0x7000_0000_0000          # "device" pointers: never dereferenced on the host
BIG = 1 << 40


def layers(H, B, dis_num_cls):
    """(name, N, Hi, I, Ho, O, k, stride, pad, pad_mode) of every convolution of SingleGenerator(3, 64, 2, 2, 6),
    SingleDiscriminator_solo_multi(3, 64, 2, dis_num_cls) and Encoder(3, 8, 64, 4), transposed convolutions as
    the convolution whose input gradient they are; every batch the trainer launches them with (B and the concatenated 2B)."""
    out = []
    for N in (B, 2 * B):
        out += [("G.down0", N, H, 3, H, 64, 7, 1, 3, 0), ("G.down1", N, H, 64, H // 2, 128, 4, 2, 1, 0),
                ("G.down2", N, H // 2, 128, H // 4, 256, 4, 2, 1, 0), ("G.res", N, H // 4, 256, H // 4, 256, 3, 1, 1, 0),
                ("G.up0", N, H // 2, 128, H // 4, 256, 4, 2, 1, 0), ("G.up1", N, H, 64, H // 2, 128, 4, 2, 1, 0),
                ("G.out", N, H, 64, H, 3, 7, 1, 3, 0)]
        for scale, (h, w0) in enumerate(((H, 64), (H // 2, 32))):
            c, hh = 3, h
            for i in range(dis_num_cls):
                o = min(w0 * 2 ** i, w0 * 8)
                out.append((f"D{scale}.c{i}", N, hh, c, hh // 2, o, 4, 2, 1, 0))
                c, hh = o, hh // 2
            out.append((f"D{scale}.last", N, hh, c, hh - 1, 1, 4, 1, 1, 0))
            out.append((f"D{scale}.cls", N, hh, c, 1, 4, hh, 1, 0, 0))
        he = (H + 2 - 7) // 2 + 1
        out.append(("E.first", N, H, 3, he, 64, 7, 2, 1, 0))
        c = 64
        for b in range(4):
            out += [(f"E.l{b}.conv1", N, he, c, he, c, 3, 1, 1, 1), (f"E.l{b}.cmp", N, he, c, he, 2 * c, 3, 1, 1, 1),
                    (f"E.l{b}.shortcut", N, he // 2, c, he // 2, 2 * c, 1, 1, 0, 0)]
            c, he = 2 * c, he // 2
    return out


def main():
    lib_path, H, B, ncls = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    os.environ["SRGAN_HIP_LIB"] = lib_path
    _lib.LIB_PATH = lib_path
    lib = _lib.load()
    mark = ctypes.CDLL(None).srgan_shim_mark
    mark.argtypes = [ctypes.c_char_p]
    p = ctypes.c_void_p(FAKE)
    for mode in (0, 1):
        assert lib.srgan_set_compute_mode(mode) == 0
        for (name, N, Hi, I, Ho, O, k, s, pad, pm) in layers(H, B, ncls):
            d = _lib.ConvDesc(N, Hi, Hi, I, Ho, Ho, O, k, k, s, pad, pm, I * k * k, k * k, k, 1)
            dref = ctypes.byref(d)
            tag = f"mode{mode} {name} N{N}"
            def call(what, rc):
                assert rc == 0, (tag, what, lib.srgan_last_error())
            mark(f"{tag} fwd".encode())
            call("fwd", lib.srgan_conv2d_fwd(dref, p, p, None, p, 2, 0.2, p, BIG, None))
            if I > 3:
                mark(f"{tag} dgrad".encode())
                call("dgrad", lib.srgan_conv2d_dgrad(dref, p, p, p, p, BIG, None))
            mark(f"{tag} wgrad".encode())
            call("wgrad", lib.srgan_conv2d_wgrad(dref, p, p, p, None, p, BIG, None))
            if mode == 1 and (lib.srgan_halo16_applicable(dref) or lib.srgan_halo16s2_applicable(dref)):
                for kind in (0, 1):
                    for i16 in (0, 1):
                        for o16 in (0, 1):
                            mark(f"{tag} halo16 kind{kind} in16={i16} out16={o16}".encode())
                            call("halo16", lib.srgan_halo16_conv(dref, kind, p, i16, p, None, p, o16, None))
                for x16 in (0, 1):
                    for d16 in (0, 1):
                        if x16 and not d16:
                            continue
                        mark(f"{tag} halo16 wgrad x16={x16} d16={d16}".encode())
                        call("halo16_wgrad", lib.srgan_halo16_wgrad(dref, p, x16, p, d16, p, p, BIG, None))
            if mode == 1 and lib.srgan_igemm16_io_applicable(dref, 0):
                for kind in (0, 1):
                    for i16 in (0, 1):
                        for o16 in (0, 1):
                            mark(f"{tag} igemm16 kind{kind} in16={i16} out16={o16}".encode())
                            call("igemm16", lib.srgan_igemm16_conv(dref, kind, p, i16, p, None, p, o16, 0, 0.0, p, BIG, None))
                mark(f"{tag} igemm16 wgrad".encode())
                call("igemm16_wgrad", lib.srgan_igemm16_wgrad(dref, p, p, p, p, BIG, None))
            # round 6: conv + activation with 16-bit tensors (the discriminator trunks' 4x4 / stride-2 layers; the generator's 7x7
            # RGB layers, whose 3-channel side is always fp32)
            for act in (0, 2):
                if mode != 1 or not lib.srgan_conv2d_io_applicable(dref, act):
                    continue
                ins = (0,) if I == 3 else (0, 1)
                outs = (0,) if O == 3 else (0, 1)
                for i16 in ins:
                    for o16 in outs:
                        mark(f"{tag} io act{act} fwd in16={i16} out16={o16}".encode())
                        call("io_fwd", lib.srgan_conv2d_io_fwd(dref, p, i16, p, None, p, o16, act, 0.01, p, BIG, None))
                        mark(f"{tag} io act{act} dgrad dy16={o16} dx16={i16}".encode())
                        call("io_dgrad", lib.srgan_conv2d_io_dgrad(dref, p, o16, p, p, i16, p, BIG, None))
                        if k == 7:
                            mark(f"{tag} io wgrad x16={i16} d16={o16}".encode())
                            call("io_wgrad", lib.srgan_halo16_wgrad(dref, p, i16, p, o16, p, p, BIG, None))
        # the instance norms of G (CBIN: 64 @ H, 128 @ H/2, 256 @ H/4) and E (64 @ he ... 512 @ he/8)
        he = (H + 2 - 7) // 2 + 1
        shapes = [(H * H, 64), (H * H // 4, 128), (H * H // 16, 256)] + [((he >> b) ** 2, 64 << b) for b in range(4)]
        for N in (B, 2 * B):
            for HW, C in shapes:
                tag = f"mode{mode} norm N{N} HW{HW} C{C}"
                ws = lib.srgan_instnorm_workspace(N, HW, C)
                mark(f"{tag} fwd".encode())
                rc = lib.srgan_instnorm_fwd(p, p, p, None, p, p, p, N, HW, C, 1e-5, 1, 0.0, p, ws, None)
                assert rc == 0, (tag, lib.srgan_last_error())
                mark(f"{tag} bwd".encode())
                rc = lib.srgan_instnorm_bwd(p, p, p, p, p, p, p, p, p, N, HW, C, 1, 0.0, p, ws, None)
                assert rc == 0, (tag, lib.srgan_last_error())
    lib.srgan_set_compute_mode(0)


if __name__ == "__main__":
    main()

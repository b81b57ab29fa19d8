"""Test infrastructure: the gfx950 code objects inside libsrgan_hip.so -- kernel descriptors (LDS, scratch, registers) and ISA.

hipcc embeds one clang offload bundle per translation unit in the `.hip_fatbin` section; every bundle holds one AMDGPU ELF for
gfx950.  `code_objects()` cuts them out, `kernel_descriptors()` reads the `amdhsa.kernels` metadata note of each with
`llvm-readelf --notes`, `disassemble()` runs `llvm-objdump -d`.  CPU only (LLVM tools of /opt/rocm)."""
import os
import re
import struct
import subprocess
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _section(path, name):
    """(offset, size) of an ELF64 section of the host library."""
    blob = open(path, "rb").read()
    assert blob[:4] == b"\x7fELF" and blob[4] == 2
    shoff, = struct.unpack_from("<Q", blob, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", blob, 0x3A)
    def sh(i):
        return struct.unpack_from("<IIQQQQIIQQ", blob, shoff + i * shentsize)
    stroff = sh(shstrndx)[4]
    for i in range(shnum):
        s = sh(i)
        n = blob[stroff + s[0]: blob.index(b"\0", stroff + s[0])].decode()
        if n == name:
            return blob, s[4], s[5]
    raise KeyError(name)


def code_objects(lib_path):
    """[(index, bytes)] -- the gfx950 ELF of every bundle in .hip_fatbin."""
    blob, off, size = _section(lib_path, ".hip_fatbin")
    fat = blob[off: off + size]
    out, pos = [], 0
    while True:
        pos = fat.find(_MAGIC, pos)
        if pos < 0:
            break
        n, = struct.unpack_from("<Q", fat, pos + len(_MAGIC))
        p = pos + len(_MAGIC) + 8
        for _ in range(n):
            o, s, tl = struct.unpack_from("<QQQ", fat, p)
            triple = fat[p + 24: p + 24 + tl].decode()
            p += 24 + tl
            if "gfx950" in triple and s:
                elf = fat[pos + o: pos + o + s]
                assert elf[:4] == b"\x7fELF", triple
                out.append((len(out), elf))
        pos += len(_MAGIC)
    assert out, "no gfx950 code object found in " + lib_path
    return out


def _with_file(elf, fn):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(elf)
        f.flush()
        return fn(f.name)


def kernel_descriptors(lib_path):
    """[{name, lds, scratch, vgpr, sgpr, agpr, spill_vgpr, spill_sgpr, max_wg, kernarg, dynamic_stack}] over all code objects."""
    ks = []
    for _, elf in code_objects(lib_path):
        txt = _with_file(elf, lambda p: subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", p], capture_output=True,
                                                       text=True, check=True).stdout)
        for block in re.split(r"\n\s*- \.agpr_count:", "\n" + txt)[1:]:
            block = ".agpr_count:" + block
            def g(key, default=0, conv=int):
                m = re.search(r"\.%s:\s*(\S+)" % re.escape(key), block)
                if not m:
                    return default
                v = m.group(1).strip("'\"")
                return conv(v) if conv is not int else int(v, 0)
            ks.append({
                "name": g("name", "", str), "lds": g("group_segment_fixed_size"), "scratch": g("private_segment_fixed_size"),
                "vgpr": g("vgpr_count"), "sgpr": g("sgpr_count"), "agpr": g("agpr_count"),
                "spill_vgpr": g("vgpr_spill_count"), "spill_sgpr": g("sgpr_spill_count"),
                "max_wg": g("max_flat_workgroup_size"), "kernarg": g("kernarg_segment_size"),
                "dynamic_stack": g("uses_dynamic_stack", "false", str) == "true",
            })
    assert ks
    return ks


def disassemble(lib_path):
    """ISA text of every code object, concatenated (llvm-objdump -d)."""
    return "\n".join(_with_file(elf, lambda p: subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", p],
                                                               capture_output=True, text=True, check=True).stdout)
                     for _, elf in code_objects(lib_path))


def demangle(names):
    try:
        r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    except FileNotFoundError:
        return list(names)
    return r.stdout.split("\n")[:len(names)] if r.returncode == 0 else list(names)

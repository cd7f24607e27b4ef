"""The register / LDS budget the co-resident pipeline rests on, read from the built library (CPU, no GPU needed).

DESIGN_HISTORY.md 4.0: a contraction workgroup (2 waves per SIMD) and the ring pre-pass share a CU only if
2 * vgpr(gram_kbits_kernel) + waves * vgpr(ring) <= 512 per SIMD (allocation granule 8) and the LDS adds up to <= 160 KiB.
A compiler or source change that pushes the contraction back to 256 VGPRs would silently turn the pipeline into the serial
order -- nothing fails, it just gets slower.  This test reads `.vgpr_count` / `.group_segment_fixed_size` of the kernels from
the code objects inside spark-examples_amd/libpcoa_hip.so."""
import os
import re
import shutil
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "spark-examples_amd", "libpcoa_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"


def _kernels(tmp_path):
    objcopy, readelf = os.path.join(LLVM, "llvm-objcopy"), os.path.join(LLVM, "llvm-readelf")
    if not (os.path.exists(LIB) and os.path.exists(objcopy) and os.path.exists(readelf)):
        pytest.skip("needs the built library and the ROCm llvm tools")
    fat = str(tmp_path / "fat.bin")
    subprocess.run([objcopy, "--dump-section", ".hip_fatbin=" + fat, LIB, os.devnull], check=True)
    data = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out = {}
    for m in re.finditer(re.escape(magic), data):
        base = m.start()
        (n,) = struct.unpack_from("<Q", data, base + 24)
        off = base + 32
        for _ in range(n):
            o, s, t = struct.unpack_from("<QQQ", data, off)
            off += 24
            triple = data[off:off + t].decode()
            off += t
            if "gfx950" not in triple or s == 0:
                continue
            co = str(tmp_path / ("co_%d.elf" % base))
            open(co, "wb").write(data[base + o:base + o + s])
            notes = subprocess.run([readelf, "--notes", co], stdout=subprocess.PIPE, universal_newlines=True, check=True).stdout
            for blk in notes.split("- .agpr_count:")[1:]:
                name = re.search(r"\.name:\s+(\S+)", blk)
                vg = re.search(r"\.vgpr_count:\s+(\d+)", blk)
                lds = re.search(r"\.group_segment_fixed_size:\s+(\d+)", blk)
                if name and vg:
                    out[name.group(1)] = (int(vg.group(1)), int(lds.group(1)) if lds else 0)
    assert out, "no gfx950 kernels found in " + LIB
    return out


def _find(kernels, needle):
    hits = {k: v for k, v in kernels.items() if needle in k}
    assert hits, "no kernel matching %r" % needle
    return hits


def test_contraction_and_ring_pre_passes_fit_one_cu_together(tmp_path):
    k = _kernels(tmp_path)
    gran = lambda v: (v + 7) // 8 * 8                       # VGPR allocation granule (wave64, gfx950)
    gram = _find(k, "gram_kbits_kernelILi3ELi2ELi2E")
    (gv, glds), = gram.values()
    assert gv <= 224, "the k-bits contraction is held to 224 VGPRs per wave (amdgpu_num_vgpr): %d" % gv
    free = 512 - 2 * gran(gv)                                # what two contraction waves leave of a SIMD's register file
    ring = _find(k, "pack_kbits_ring_kernelILi8ELi0ELi0E")  # the fp32 ring pre-pass the library launches (R = 8, default policy)
    (rv, _), = ring.values()
    assert 2 * gran(rv) <= free, "two ring waves per SIMD (two workgroups per CU) must fit: 2 x %d > %d" % (gran(rv), free)
    u8 = _find(k, "pack_u8_kbits_ring_kernelILi8ELi0E")
    (uv, _), = u8.values()
    assert gran(uv) <= free, "one uint8 ring wave per SIMD must fit: %d > %d" % (gran(uv), free)
    tr = _find(k, "transpose_bits_kbits_kernel")
    assert all(gran(v) <= free for v, _ in tr.values()), "the bitset transpose must fit one wave per SIMD: %r" % tr
    # LDS: contraction ring (3 stages of 8 KiB) + two fp32 ring workgroups (32 KiB each, dynamic) <= 160 KiB
    assert glds + 2 * 32 * 1024 <= 160 * 1024, glds

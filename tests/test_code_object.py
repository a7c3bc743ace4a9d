"""CPU-side checks of the built library's device code (no GPU needed): the shared library carries gfx950 code objects
only, every kernel's register / LDS budget is inside the CU's limits and no kernel spills registers to scratch
memory.  Parses the clang offload bundles and the AMDGPU metadata notes of libyolov3_hip.so directly."""
import os
import struct

import msgpack
import pytest

from golden_util import ROOT

LIB = os.path.join(ROOT, "pytorch-yolov3_amd", "lib", "libyolov3_hip.so")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _code_objects(data):
    """(target triple, ELF bytes) of every device entry of every offload bundle in the file."""
    pos = 0
    while True:
        i = data.find(MAGIC, pos)
        if i < 0:
            return
        nent = struct.unpack_from("<Q", data, i + 24)[0]
        off = i + 32
        for _ in range(nent):
            o, sz, tl = struct.unpack_from("<QQQ", data, off)
            triple = data[off + 24:off + 24 + tl].decode()
            off += 24 + tl
            if sz and not triple.startswith("host"):
                yield triple, data[i + o:i + o + sz]
        pos = i + 24


def _kernels(elf):
    """amdhsa.kernels entries of one code object (NT_AMDGPU_METADATA note, msgpack)."""
    shoff = struct.unpack_from("<Q", elf, 0x28)[0]
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    for k in range(shnum):
        sh = elf[shoff + k * shentsize:shoff + (k + 1) * shentsize]
        if struct.unpack_from("<I", sh, 4)[0] != 7:          # SHT_NOTE
            continue
        so, ss = struct.unpack_from("<QQ", sh, 0x18)
        p = so
        while p < so + ss:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            d0 = p + 12 + ((namesz + 3) & ~3)
            if ntype == 32 and elf[p + 12:p + 12 + namesz].startswith(b"AMDGPU"):
                md = msgpack.unpackb(elf[d0:d0 + descsz], raw=False, strict_map_key=False)
                for kern in md.get("amdhsa.kernels", []):
                    yield kern
            p = d0 + ((descsz + 3) & ~3)


@pytest.fixture(scope="module")
def kernels():
    if not os.path.exists(LIB):
        pytest.skip("library not built (python -c 'import __graft_entry__ as g; g.build()')")
    data = open(LIB, "rb").read()
    objs = list(_code_objects(data))
    assert objs, "no device code in the library"
    out = []
    for triple, elf in objs:
        assert "gfx950" in triple, triple                    # one target, no multi-arch fallbacks
        out.extend(_kernels(elf))
    return out


def test_library_has_the_hot_path_kernels(kernels):
    names = " ".join(k[".name"] for k in kernels)
    for needle in ("conv_halo_ws_kernel", "conv_patch_wsp_kernel", "conv_igemm2_kernel", "conv_igemm3_kernel",
                   "conv_stem_s2_fused_kernel", "conv_resblock_fused_kernel", "yolo_decode_kernel", "detect_kernel",
                   "pack_records_kernel", "resize"):
        assert needle in names, needle


def test_no_kernel_spills_and_budgets_fit_a_cu(kernels):
    for k in kernels:
        name = k[".name"]
        assert k[".private_segment_fixed_size"] == 0, (name, "uses scratch memory (register spills)")
        assert k[".vgpr_count"] + k.get(".agpr_count", 0) <= 512, name
        assert k[".group_segment_fixed_size"] <= 160 * 1024, name
        assert k[".wavefront_size"] == 64, name


def test_device_code_hash_does_not_depend_on_the_build_directory(tmp_path):
    """bench.py ties profiles/r0N_traffic.json to the library by a hash over its DEVICE code.  hipcc embeds an id derived
    from the source path in every code object's symbol tables; the hash must not see it: one source
    compiled into two directories, under two different source paths, hashes the same -- and the raw objects do differ."""
    import shutil
    import subprocess
    import sys
    sys.path.insert(0, ROOT)
    import bench
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    csrc = os.path.join(ROOT, "pytorch-yolov3_amd", "csrc")
    objs = []
    for sub in ("a", os.path.join("b", "deeper")):
        d = tmp_path / sub
        d.mkdir(parents=True)
        for name in ("yolo_decode.hip", "common.h", "decode_core.h"):
            shutil.copy(os.path.join(csrc, name), str(d / name))
        out = str(d / "yolo_decode.o")
        subprocess.check_call([hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
                               "-ffp-contract=off", "-Wno-unused-function", "-c", str(d / "yolo_decode.hip"), "-o", out])
        objs.append(out)
    raw = [open(o, "rb").read() for o in objs]
    code = [[elf for _, elf in _code_objects(r)] for r in raw]
    assert code[0] != code[1], "expected the path-derived id to differ between the two builds"
    assert bench.device_code_sha256(objs[0]) == bench.device_code_sha256(objs[1])
    # ... and it is a hash of the kernels: another source gives another value
    assert bench.device_code_sha256(objs[0]) != bench.device_code_sha256(LIB)

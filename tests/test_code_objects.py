"""Properties of the built library's gfx950 code objects, read from their metadata (no GPU needed)."""
import os


def _device_code_objects(so_path):
    """The gfx950 code objects embedded in the library: every clang offload bundle of its .hip_fatbin section."""
    import struct
    import subprocess
    import tempfile
    objcopy = "/opt/rocm/lib/llvm/bin/llvm-objcopy"
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.check_call([objcopy, "--dump-section", ".hip_fatbin=" + fat, so_path, os.path.join(td, "copy.so")])
        blob = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out = []
    pos = blob.find(magic)
    while pos >= 0:
        n, = struct.unpack_from("<Q", blob, pos + len(magic))
        q = pos + len(magic) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, q)
            triple = blob[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if "gfx950" in triple:
                out.append(blob[pos + off:pos + off + size])
        pos = blob.find(magic, pos + 1)
    return out


def test_no_kernel_uses_scratch_memory(tmp_path):
    """Every kernel of the library keeps its per-lane state in registers (or LDS): private_segment_fixed_size == 0 in each
    kernel's metadata.  A 48-byte scratch array in the matrix-core NN kernel (coordinates read back at a run-time offset) made
    correspondences change from run to run on the MI355X: arrays indexed at run time belong in LDS."""
    import re
    import subprocess
    from mm_masking_amd import _lib
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not os.path.exists(readelf):
        import pytest
        pytest.skip("llvm-readelf not available")
    objs = _device_code_objects(_lib.build())
    assert len(objs) >= 4                              # icp, radar, unet, loader (sources without kernels embed none)
    kernels = 0
    for k, elf in enumerate(objs):
        path = tmp_path / ("dev%d.elf" % k)
        path.write_bytes(elf)
        notes = subprocess.check_output([readelf, "--notes", str(path)], text=True)
        names = re.findall(r"^\s+\.name:\s+(\S+)\s*$", notes, flags=re.M)
        sizes = re.findall(r"\.private_segment_fixed_size:\s+(\d+)", notes)
        kernel_names = [n for n in names if n.startswith("_Z")]
        assert len(sizes) == len(kernel_names) and sizes, (len(sizes), len(kernel_names))
        kernels += len(sizes)
        bad = [(n, s) for n, s in zip(kernel_names, sizes) if int(s) != 0]
        assert not bad, "kernels with scratch memory: %r" % bad
        # no kernel may ask for a run-time sized stack either (round 4: the failing round-3 code object had a fixed 48-byte
        # segment and .uses_dynamic_stack false -- the descriptor was not the cause -- but a dynamic stack would defeat the
        # size check above, so it is asserted too)
        dyn = re.findall(r"\.uses_dynamic_stack:\s+(\S+)", notes)
        assert len(dyn) == len(kernel_names) and all(d == "false" for d in dyn), dyn
    assert kernels >= 60

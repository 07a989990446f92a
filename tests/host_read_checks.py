"""Checks of mmk_host_read_rows / mmk_host_read_rows_batch (include/mmk.h) against numpy, on any library that exports them:
the product's libmmk_hip.so (tests/test_loader_cpu.py) and the CPU-only sanitizer build of the same source
(``python host_read_checks.py <lib.so> <dir>`` under LD_PRELOAD=libasan.so)."""
import ctypes
import os
import sys

import numpy as np


class ReadJob(ctypes.Structure):
    """mmk_read_job of include/mmk.h."""
    _fields_ = [("path", ctypes.c_char_p), ("header_bytes", ctypes.c_int64), ("rows", ctypes.c_int32),
                ("row_bytes", ctypes.c_int32), ("col0", ctypes.c_int32), ("ncols", ctypes.c_int32), ("roll", ctypes.c_int32),
                ("reserved", ctypes.c_int32), ("dst", ctypes.c_void_p)]


def declare(L):
    i32 = ctypes.c_int32
    L.mmk_host_read_rows.restype = ctypes.c_int
    L.mmk_host_read_rows.argtypes = [ctypes.c_char_p, ctypes.c_int64, i32, i32, i32, i32, i32, ctypes.c_void_p]
    L.mmk_host_read_rows_batch.restype = ctypes.c_int
    L.mmk_host_read_rows_batch.argtypes = [ctypes.POINTER(ReadJob), i32, i32]
    L.mmk_last_error.restype = ctypes.c_char_p
    L.mmk_last_error.argtypes = []
    return L


def run(L, tmp, ReadJob=ReadJob):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (7, 23), dtype=np.uint8)
    path = os.path.join(tmp, "rows.u8")
    with open(path, "wb") as f:
        f.write(b"HEAD1234")
        f.write(img.tobytes())
    for roll in (0, 3, -2, 7, 9):
        out = np.zeros((7, 10), np.uint8)
        assert L.mmk_host_read_rows(path.encode(), 8, 7, 23, 5, 10, roll, out.ctypes.data) == 0, L.mmk_last_error()
        assert np.array_equal(out, np.roll(img[:, 5:15], roll, axis=0)), roll
    full = np.zeros((7, 23), np.uint8)
    assert L.mmk_host_read_rows(path.encode(), 8, 7, 23, 0, 23, 0, full.ctypes.data) == 0
    assert np.array_equal(full, img)
    assert L.mmk_host_read_rows(path.encode(), 8, 8, 23, 0, 23, 0, full.ctypes.data) != 0 and b"shorter" in L.mmk_last_error()
    assert L.mmk_host_read_rows(os.path.join(tmp, "missing").encode(), 0, 1, 4, 0, 4, 0, full.ctypes.data) != 0
    # the batched form: 12 jobs on 3 threads, then one bad job among them
    outs = [np.zeros((7, 10), np.uint8) for _ in range(12)]
    jobs = (ReadJob * 12)()
    pb = path.encode()
    for k, o in enumerate(outs):
        jobs[k].path, jobs[k].header_bytes, jobs[k].rows, jobs[k].row_bytes = pb, 8, 7, 23
        jobs[k].col0, jobs[k].ncols, jobs[k].roll, jobs[k].dst = k, 10, k - 4, o.ctypes.data
    assert L.mmk_host_read_rows_batch(jobs, 12, 3) == 0, L.mmk_last_error()
    for k, o in enumerate(outs):
        assert np.array_equal(o, np.roll(img[:, k:k + 10], k - 4, axis=0)), k
    jobs[5].rows = 9
    assert L.mmk_host_read_rows_batch(jobs, 12, 3) != 0 and b"1 of 12 jobs failed" in L.mmk_last_error()
    # Navtech-sized rows through the staging block: 400 rows of 3 371 bytes, the 3 360 power bytes cut out, rolled
    nav = rng.integers(0, 256, (400, 3371), dtype=np.uint8)
    npath = os.path.join(tmp, "navtech.u8")
    nav.tofile(npath)
    cut = np.zeros((400, 3360), np.uint8)
    assert L.mmk_host_read_rows(npath.encode(), 0, 400, 3371, 11, 3360, 37, cut.ctypes.data) == 0, L.mmk_last_error()
    assert np.array_equal(cut, np.roll(nav[:, 11:], 37, axis=0))
    enc = np.zeros((400, 2), np.uint8)                      # the narrow cut (one pread per row)
    assert L.mmk_host_read_rows(npath.encode(), 0, 400, 3371, 8, 2, -5, enc.ctypes.data) == 0
    assert np.array_equal(enc, np.roll(nav[:, 8:10], -5, axis=0))
    # rows longer than the staging block (these used to be refused): a wide cut of 300 000-byte rows, and rows that
    # just fit / just exceed the block
    for rb in (300000, 65536, 65537):
        wide = rng.integers(0, 256, (3, rb), dtype=np.uint8)
        wpath = os.path.join(tmp, "wide_%d.u8" % rb)
        wide.tofile(wpath)
        got = np.zeros((3, rb - 100), np.uint8)
        assert L.mmk_host_read_rows(wpath.encode(), 0, 3, rb, 60, rb - 100, 1, got.ctypes.data) == 0, L.mmk_last_error()
        assert np.array_equal(got, np.roll(wide[:, 60:rb - 40], 1, axis=0)), rb
        os.remove(wpath)


if __name__ == "__main__":
    run(declare(ctypes.CDLL(sys.argv[1])), sys.argv[2])
    print("host_read_checks ok")

"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/press_hip.h declares.  No compute calls (no GPU here)."""
import os
import re

import pytest

from honours_amd import build, press

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build.build()
    return press.load_library()


def _declared_symbols():
    """function names declared in include/press_hip.h (incl. the macro-free drop-in list)"""
    src = open(os.path.join(ROOT, "include", "press_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set(re.findall(r"\b([a-z_][a-z0-9_]*)\s*\([^;{]*\)\s*;", src))
    return {n for n in names if n not in ("defined",)}


def test_header_symbols_exported(lib):
    decl = _declared_symbols()
    assert len(decl) >= 60
    missing = [s for s in sorted(decl) if not hasattr(lib, s)]
    assert not missing, missing
    # and the python mirror knows the same list
    assert set(press.HEADER_SYMBOLS) == decl


def test_bounds_match_oracle(lib, oracle):
    """X_bound is host arithmetic: same values as the oracle"""
    for m in press.METHODS:
        for n in (1, 2, 7, 8, 9, 1000, 7329, 155185, 200000, 5724000):
            assert press.bound(m, n) == oracle.bound(m, n), (m, n)
            assert lib.press_hip_bound(press.METHODS[m], n) == oracle.bound(m, n), (m, n)


def test_huffman_objects_roundtrip(lib, oracle):
    """read_code_table + build_symbol_encoder (press/test.c:3786-3791) reproduce the codes"""
    import ctypes

    t = press.HuffmanTable()
    want = oracle.table()
    se = ctypes.cast(t.se, ctypes.POINTER(ctypes.c_void_p * 256)).contents

    class Code(ctypes.Structure):
        _fields_ = [("numbits", ctypes.c_ulong), ("bits", ctypes.POINTER(ctypes.c_ubyte))]

    for s in range(256):
        c = ctypes.cast(se[s], ctypes.POINTER(Code)).contents
        bits = 0
        for k in range(c.numbits):
            if c.bits[k // 8] & (1 << (k % 8)):
                bits |= 1 << k
        assert (c.numbits, bits) == want[s], s
    t.close()


def test_no_gpu_fails_loudly(lib):
    """without a device the product path reports an error - it never falls back to the CPU"""
    import numpy as np
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    ret, out = press.press("hasgam_vbsse21_zdq", np.arange(100, dtype=np.int16))
    assert ret != 0 and out == b""
    assert "HIP" in press.last_error() or "hip" in press.last_error()


def test_scratch_registry_without_gpu():
    """every scratch buffer of the library is on the list press_hip_shutdown() walks (a buffer cannot be
    forgotten: it links itself in when it is constructed); nothing is allocated before the first call"""
    import ctypes

    lib = press.load_library()
    lib.press_hip_scratch_buffers.restype = ctypes.c_uint32
    lib.press_hip_scratch_buffers.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
    nbytes = ctypes.c_uint64(1)
    nbuf = lib.press_hip_scratch_buffers(ctypes.byref(nbytes))
    # as many as the context declares (DevBuf a, b, c; lines of struct Ctx)
    src = open(os.path.join(os.path.dirname(build.CSRC), "csrc", "press_abi.hip")).read()
    declared = sum(len(l.split(";")[0].split(",")) for l in re.findall(r"^\tDevBuf ([a-z][^;]*;)", src, re.M))
    assert nbuf == declared >= 40, (nbuf, declared)
    import torch
    if not torch.cuda.is_available():
        assert nbytes.value == 0
    lib.press_hip_shutdown()  # harmless without a context


def test_one_libzstd_per_process():
    """round 2's abort under rocprofv3 ("munmap_chunk(): invalid pointer"): the profiler's tool library had the
    system's libzstd in the global scope and the library / bench.py opened another version by absolute path - the
    second copy's internal calls bound to the first and freed its memory.  With a libzstd preloaded the way the
    profiler does it, both openers must bind to THAT copy and compress without aborting."""
    import glob
    import subprocess
    import sys

    sys_z = [p for p in glob.glob("/usr/lib/x86_64-linux-gnu/libzstd.so.1") + glob.glob("/lib/x86_64-linux-gnu/libzstd.so.1")]
    if not sys_z:
        pytest.skip("no system libzstd to preload")
    code = (
        "import ctypes, numpy as np\n"
        "from honours_amd import press\n"
        "z = press.open_libzstd()\n"
        "src = np.tile(np.arange(50, dtype=np.uint8), 4000); dst = np.zeros(src.size + 1024, dtype=np.uint8)\n"
        "c = z.ZSTD_compress(dst.ctypes.data, dst.size, src.ctypes.data, src.size, 1)\n"
        "assert not z.ZSTD_isError(c) and 0 < c < src.size\n"
        "lib = press.load_library()\n"
        "assert press.bound('zstd_svb_zd', 100000) > 100000\n"   # the library's own dlopen (zstd_open)
        "c = z.ZSTD_compress(dst.ctypes.data, dst.size, src.ctypes.data, src.size, 1)\n"
        "assert not z.ZSTD_isError(c)\n"
        "print('ok')\n")
    env = dict(os.environ, LD_PRELOAD=sys_z[0], PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "ok" in p.stdout, (p.returncode, p.stderr[-500:])

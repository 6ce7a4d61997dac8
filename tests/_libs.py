"""ctypes loaders shared by the tests, bench.py's cpu_baseline leg and smoke().

* ``oracle()``   -> oracle/libpress_oracle.so   (our CPU restatement; checker only)
* ``reference()``-> oracle/_ref/libpress_ref.so (the real reference, when built; checker only)

Nothing here is imported by the product package ``honours_amd``.
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
TABLE = os.path.join(ROOT, "honours_amd", "data", "NA12878_zd.huffman")

# oracle/press_methods.h
METHODS = {
    "svb12": 0, "svb12_zd": 1, "svb_zd": 2, "zstd_svb_zd": 3, "zstd_svb12_zd": 4,
    "vbe21_zd": 5, "vbbe21_zd": 6, "vbsbe21_zd": 7, "vbsse21_zd": 8,
    "shuffman_vbe21_zd": 9, "shuffman_vbbe21_zd": 10, "shuffman_vbsbe21_zd": 11,
    "shuffman_vbsse21_zd": 12, "hasgam_vbsse21_zdq": 13, "zstd_hasgam_vbsse21_zdq": 14,
    "slow5_svb_zd": 15,  # BLOW5's signal codec (slow5lib svb-zd), SURVEY 8f-2
    "rc_vbe21_zd": 16,   # vbe21 + TurboRC order-0 range coder, SURVEY 8f-1
    "rcc_vbe21_zd": 17,  # vbe21 + TurboRC order-1 range coder, SURVEY 8f-1
    "rccm_vbbe21_zd": 18,  # vbbe21 + TurboRC order 1-0 context mixing + SSE, SURVEY 8f-4
}
DETERMINISTIC = [m for m in METHODS if not m.startswith("zstd_")]
RC_FAMILY = ("rc_vbe21_zd", "rcc_vbe21_zd", "rccm_vbbe21_zd")


def rc_stored_raw(method, stream, n):
    """TurboRC stores tiny / incompressible inputs raw (rcutil_.h:161: once the coder's output reaches
    n*255/256 - 8 bytes) and nothing tells its decoder: such streams are outside the reference's lossless
    domain (its own decoder may even abort on them).  True if `stream` of an n-sample read is one."""
    nex = int.from_bytes(stream[2:6], "little")
    if method == "rccm_vbbe21_zd":  # vbbe21 section (press.c:6931-6940)
        sec = 4
        if nex > 1:
            lp = int.from_bytes(stream[2 + sec:6 + sec], "little")
            sec += 4 + lp
            lv = int.from_bytes(stream[2 + sec:6 + sec], "little")
            sec += 4 + lv
        elif nex == 1:
            sec += 6
    else:                           # vbe21 section: u32 positions, u16 values
        sec = 4 + 6 * nex
    nlow = n - 1 - nex
    return nlow > 0 and len(stream) - 2 - sec == nlow

_u8p = ctypes.POINTER(ctypes.c_uint8)


def fnv1a32(b: bytes) -> int:
    h = 0x811C9DC5
    for x in b:
        h = ((h ^ x) * 0x01000193) & 0xFFFFFFFF
    return h


def _build_oracle():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "oracle"], check=True)


class _Codec:
    """Uniform press/depress/bound over a library exporting <pfx>_bound/_press/_depress."""

    def __init__(self, lib, pfx):
        self.lib = lib
        self.pfx = pfx
        self._bound = getattr(lib, pfx + "_bound")
        self._bound.restype = ctypes.c_uint64
        self._bound.argtypes = [ctypes.c_int, ctypes.c_uint32]
        self._press = getattr(lib, pfx + "_press")
        self._press.restype = ctypes.c_int
        self._press.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p,
                                ctypes.POINTER(ctypes.c_uint64)]
        self._depress = getattr(lib, pfx + "_depress")
        self._depress.restype = ctypes.c_int
        self._depress.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32,
                                  ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32)]
        self._load = getattr(lib, pfx + "_load_table")
        self._load.restype = ctypes.c_int
        self._load.argtypes = [ctypes.c_char_p]
        self._time = getattr(lib, pfx + "_time_batch")
        self._time.restype = ctypes.c_int
        self._time.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32,
                               ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                               ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
        self._code = getattr(lib, pfx + "_table_code")
        self._code.restype = ctypes.c_int
        self._code.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint64)]

    def load_table(self, path=TABLE):
        if self._load(path.encode()) != 0:
            raise RuntimeError("cannot load Huffman table " + path)

    def table(self):
        out = []
        for s in range(256):
            ln, bits = ctypes.c_uint32(), ctypes.c_uint64()
            if self._code(s, ctypes.byref(ln), ctypes.byref(bits)) != 0:
                raise RuntimeError("no code for symbol %d" % s)
            out.append((ln.value, bits.value))
        return out

    def bound(self, method, n):
        return self._bound(METHODS[method] if isinstance(method, str) else method, n)

    def press(self, method, sig, cap=None):
        """-> (ret, bytes)"""
        m = METHODS[method] if isinstance(method, str) else method
        sig = np.ascontiguousarray(sig, dtype=np.int16)
        n = sig.size
        # X_bound is a heuristic that is too small for tiny or exception-heavy reads
        # (press.c:2575; n < 4 always) - the helpers give slack unless a test asks otherwise
        cap = int(self.bound(m, n)) + 1024 if cap is None else cap
        out = np.zeros(cap + 64, dtype=np.uint8)
        nout = ctypes.c_uint64(cap)
        ret = self._press(m, sig.ctypes.data, n, out.ctypes.data, ctypes.byref(nout))
        return ret, (out[: nout.value].tobytes() if ret == 0 else b"")

    def depress(self, method, comp: bytes, n):
        """-> (ret, int16 array)"""
        m = METHODS[method] if isinstance(method, str) else method
        buf = np.frombuffer(comp + b"\0" * 64, dtype=np.uint8).copy()
        out = np.zeros(n + 64, dtype=np.int16)
        nout = ctypes.c_uint32(n)
        ret = self._depress(m, buf.ctypes.data, len(comp), n, out.ctypes.data, ctypes.byref(nout))
        return ret, (out[: nout.value].copy() if ret == 0 else out[:0])

    def time_batch(self, method, sig, off, check=True):
        m = METHODS[method] if isinstance(method, str) else method
        sig = np.ascontiguousarray(sig, dtype=np.int16)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        ps, ds, pb = ctypes.c_double(), ctypes.c_double(), ctypes.c_uint64()
        ret = self._time(m, sig.ctypes.data, off.ctypes.data, off.size - 1, ctypes.byref(ps),
                         ctypes.byref(ds), ctypes.byref(pb), int(check))
        if ret != 0:
            raise RuntimeError("time_batch failed: %d" % ret)
        return ps.value, ds.value, pb.value


_cache = {}


def oracle() -> _Codec:
    if "o" not in _cache:
        path = os.path.join(ORACLE_DIR, "libpress_oracle.so")
        src = os.path.join(ORACLE_DIR, "press_oracle.c")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            _build_oracle()
        lib = ctypes.CDLL(path)
        c = _Codec(lib, "po")
        c.load_table()
        # primitives
        lib.po_svb32_encode.restype = ctypes.c_uint64
        lib.po_svb32_encode.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
        lib.po_svb32_decode.restype = ctypes.c_uint64
        lib.po_svb32_decode.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
        lib.po_uint_pack.restype = ctypes.c_uint64
        lib.po_uint_pack.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p]
        lib.po_uint_unpack.restype = ctypes.c_uint64
        lib.po_uint_unpack.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p]
        lib.po_zigdelta_u16.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
        lib.po_unzigdelta_u16.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
        _cache["o"] = c
    return _cache["o"]


def have_reference() -> bool:
    return os.path.exists(os.path.join(ORACLE_DIR, "_ref", "libpress_ref.so"))


def reference() -> _Codec:
    if "r" not in _cache:
        lib = ctypes.CDLL(os.path.join(ORACLE_DIR, "_ref", "libpress_ref.so"))
        c = _Codec(lib, "ref")
        c.load_table()
        _cache["r"] = c
    return _cache["r"]

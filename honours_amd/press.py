"""Host-side mirror of the reference's per-method interface (press/press.h) over
libpress_hip.so, plus the batch entry points.

Per-read calls go through the DROP-IN C symbols (``svb12_zd_press`` ...) exactly as
press/test.c's ``test_X`` functions call them (test.c:1756-1815): bound -> press ->
depress.  Batch calls take torch CUDA tensors (device memory plumbing only) and enqueue
on torch's current stream.

No fallback: if the library is missing or no GPU is present, calls raise.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PRESS_HIP_LIB", os.path.join(_HERE, "libpress_hip.so"))  # override: diagnostic builds
TABLE_PATH = os.path.join(_HERE, "data", "NA12878_zd.huffman")

# include/press_hip.h enum press_hip_method
METHODS = {
    "svb12": 0, "svb12_zd": 1, "svb_zd": 2, "zstd_svb_zd": 3, "zstd_svb12_zd": 4,
    "vbe21_zd": 5, "vbbe21_zd": 6, "vbsbe21_zd": 7, "vbsse21_zd": 8,
    "shuffman_vbe21_zd": 9, "shuffman_vbbe21_zd": 10, "shuffman_vbsbe21_zd": 11,
    "shuffman_vbsse21_zd": 12, "hasgam_vbsse21_zdq": 13, "zstd_hasgam_vbsse21_zdq": 14,
    "slow5_svb_zd": 15, "rc_vbe21_zd": 16, "rcc_vbe21_zd": 17, "rccm_vbbe21_zd": 18,
}
BATCH_METHODS = [m for m in METHODS if not m.startswith("zstd_")]
FAILED = (1 << 64) - 1

# method -> (bound symbol, press symbol, depress symbol, calling convention)
#   "svb"    void press(in, n, out, &nout64);  void depress(in, nsamples, out, &nout64)
#   "svb_nz" as svb, depress without nout (svb12_depress)
#   "vb"     void press(in, n32, out, &nout64); void depress(in, nbytes, out, &nout32)
#   "int"    int press(...) ; int depress(in, nbytes, out, &nout32)
#   "shuff"  int press(se, ...); int depress(root, ...)
_SYMS = {
    "svb12": ("svb12_bound", "svb12_press", "svb12_depress", "svb_nz"),
    "svb12_zd": ("svb12_zd_bound", "svb12_zd_press", "svb12_zd_depress", "svb"),
    "svb_zd": ("svb_zd_bound_16", "svb_zd_press_16", "svb_zd_depress_16", "svb"),
    "zstd_svb_zd": ("zstd_svb_zd_bound_16", "zstd_svb_zd_press_16", "zstd_svb_zd_depress_16", "int"),
    "zstd_svb12_zd": ("zstd_svb12_zd_bound", "zstd_svb12_zd_press", "zstd_svb12_zd_depress", "int"),
    "vbe21_zd": ("vbe21_zd_bound_16", "vbe21_zd_press_16", "vbe21_zd_depress_16", "vb"),
    "vbbe21_zd": ("vbbe21_zd_bound_16", "vbbe21_zd_press_16", "vbbe21_zd_depress_16", "vb"),
    "vbsbe21_zd": ("vbsbe21_zd_bound_16", "vbsbe21_zd_press_16", "vbsbe21_zd_depress_16", "vb"),
    "vbsse21_zd": ("vbsse21_zd_bound_16", "vbsse21_zd_press_16", "vbsse21_zd_depress_16", "vb"),
    "shuffman_vbe21_zd": ("shuffman_vbe21_zd_bound_16", "shuffman_vbe21_zd_press_16",
                          "shuffman_vbe21_zd_depress_16", "shuff"),
    "shuffman_vbbe21_zd": ("shuffman_vbbe21_zd_bound_16", "shuffman_vbbe21_zd_press_16",
                           "shuffman_vbbe21_zd_depress_16", "shuff"),
    "shuffman_vbsbe21_zd": ("shuffman_vbsbe21_zd_bound_16", "shuffman_vbsbe21_zd_press_16",
                            "shuffman_vbsbe21_zd_depress_16", "shuff"),
    "shuffman_vbsse21_zd": ("shuffman_vbsse21_zd_bound_16", "shuffman_vbsse21_zd_press_16",
                            "shuffman_vbsse21_zd_depress_16", "shuff"),
    "hasgam_vbsse21_zdq": ("hasgam_vbsse21_zdq_bound_16", "hasgam_vbsse21_zdq_press_16",
                           "hasgam_vbsse21_zdq_depress_16", "int"),
    "zstd_hasgam_vbsse21_zdq": ("zstd_hasgam_vbsse21_zdq_bound_16", "zstd_hasgam_vbsse21_zdq_press_16",
                                "zstd_hasgam_vbsse21_zdq_depress_16", "int"),
    # BLOW5's signal codec (slow5lib svb-zd)
    "slow5_svb_zd": ("slow5_svb_zd_bound", "slow5_svb_zd_press", "slow5_svb_zd_depress", "int"),
    # vbe21 + order-0 range coder (one read per lane on the device)
    "rc_vbe21_zd": ("rc_vbe21_zd_bound_16", "rc_vbe21_zd_press_16", "rc_vbe21_zd_depress_16", "vb"),
    # ... order 1 (one read per workgroup)
    "rcc_vbe21_zd": ("rcc_vbe21_zd_bound_16", "rcc_vbe21_zd_press_16", "rcc_vbe21_zd_depress_16", "vb"),
    "rccm_vbbe21_zd": ("rccm_vbbe21_zd_bound_16", "rccm_vbbe21_zd_press_16", "rccm_vbbe21_zd_depress_16", "vb"),
}

# every symbol include/press_hip.h declares (checked by tests/test_abi_symbols.py)
HEADER_SYMBOLS = sorted(set(
    [s for t in _SYMS.values() for s in t[:3]] +
    ["read_code_table", "build_symbol_encoder", "free_encoder", "free_huffman_tree",
     "press_hip_last_error", "press_hip_set_device", "press_hip_set_stream", "press_hip_reset_stream",
     "press_hip_get_stream",
     "press_hip_synchronize", "press_hip_load_table_file", "press_hip_set_table", "press_hip_bound",
     "press_hip_press_batch", "press_hip_depress_batch", "press_hip_workspace_bytes",
     "press_hip_kernel_timing", "press_hip_kernel_times",
     "press_hip_slow5_ptr_compress_svb_zd", "press_hip_slow5_ptr_depress_svb_zd",
     "press_hip_blow5_open", "press_hip_blow5_close", "press_hip_blow5_methods", "press_hip_blow5_next",
     "press_hip_blow5_last_error", "press_hip_blow5_next_records", "press_hip_blow5_create",
     "press_hip_blow5_write", "press_hip_blow5_finish", "press_hip_blow5_write_batch", "press_hip_blow5_index",
     "press_hip_blow5_threads",
     "press_hip_shutdown", "press_hip_scratch_buffers", "press_hip_host_alloc", "press_hip_host_free",
     "press_hip_zstd_host_frames"]))


class PressError(RuntimeError):
    pass


_lib = None


def load_library(path=LIB_PATH):
    """dlopen libpress_hip.so (no GPU needed for that) - raises if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(path):
            raise PressError("%s is not built: run `python -m honours_amd.build` "
                             "(there is no CPU fallback)" % path)
        _lib = ctypes.CDLL(path)
        _lib.press_hip_last_error.restype = ctypes.c_char_p
        _lib.press_hip_bound.restype = ctypes.c_uint64
        _lib.press_hip_bound.argtypes = [ctypes.c_int, ctypes.c_uint32]
        _lib.press_hip_workspace_bytes.restype = ctypes.c_uint64
        _lib.press_hip_workspace_bytes.argtypes = [ctypes.c_int, ctypes.c_uint64, ctypes.c_uint32]
        _lib.press_hip_get_stream.restype = ctypes.c_void_p
        _lib.press_hip_set_stream.argtypes = [ctypes.c_void_p]
        _lib.press_hip_load_table_file.argtypes = [ctypes.c_char_p]
        _lib.press_hip_press_batch.restype = ctypes.c_int
        _lib.press_hip_press_batch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_uint32, ctypes.c_uint64, ctypes.c_void_p,
                                               ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        _lib.press_hip_depress_batch.restype = ctypes.c_int
        _lib.press_hip_depress_batch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                 ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p,
                                                 ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p,
                                                 ctypes.c_int]
        libc = ctypes.CDLL(None)
        libc.fopen.restype = ctypes.c_void_p
        libc.fopen.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
        libc.fclose.argtypes = [ctypes.c_void_p]
        libc.malloc.restype = ctypes.c_void_p
        libc.malloc.argtypes = [ctypes.c_size_t]
        _lib._libc = libc
        _lib.read_code_table.restype = ctypes.c_bool
        _lib.read_code_table.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p),
                                         ctypes.POINTER(ctypes.c_uint)]
        _lib.build_symbol_encoder.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        _lib.free_encoder.argtypes = [ctypes.c_void_p]
        _lib.free_huffman_tree.argtypes = [ctypes.c_void_p]
    return _lib


def open_libzstd():
    """ctypes handle to a libzstd for tests and bench.py (never the product path), or None.  The copy the process
    already holds comes first: a second copy of another version frees the first one's allocations and glibc aborts
    ("free(): invalid pointer" - round 2's rocprofv3 run); an absolute path only last, bound
    to itself (RTLD_DEEPBIND)."""
    import os
    tries = [("libzstd.so.1", os.RTLD_NOW | os.RTLD_NOLOAD), ("libzstd.so.1", os.RTLD_NOW), ("libzstd.so", os.RTLD_NOW),
             ("/opt/conda/lib/libzstd.so.1", os.RTLD_NOW | os.RTLD_DEEPBIND)]
    for name, mode in tries:
        try:
            z = ctypes.CDLL(name, mode=mode)
        except OSError:
            continue
        z.ZSTD_decompress.restype = ctypes.c_size_t
        z.ZSTD_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
        z.ZSTD_compress.restype = ctypes.c_size_t
        z.ZSTD_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        z.ZSTD_isError.argtypes = [ctypes.c_size_t]
        return z
    return None


def last_error():
    return load_library().press_hip_last_error().decode()


def _mid(method):
    return METHODS[method] if isinstance(method, str) else int(method)


class HuffmanTable:
    """The caller-owned table objects of press/test.c:3786-3791: read_code_table on the
    table file, then build_symbol_encoder into a malloc'd SymbolEncoder."""

    def __init__(self, path=TABLE_PATH):
        lib = load_library()
        fp = lib._libc.fopen(path.encode(), b"r")
        if not fp:
            raise PressError("cannot open " + path)
        self.root = ctypes.c_void_p()
        nbytes = ctypes.c_uint()
        ok = lib.read_code_table(fp, ctypes.byref(self.root), ctypes.byref(nbytes))
        lib._libc.fclose(fp)
        if not ok:
            raise PressError("read_code_table failed on " + path)
        self.se = lib._libc.malloc(256 * ctypes.sizeof(ctypes.c_void_p))
        ctypes.memset(self.se, 0, 256 * ctypes.sizeof(ctypes.c_void_p))
        lib.build_symbol_encoder(self.root, self.se)

    def close(self):
        lib = load_library()
        if self.se:
            lib.free_encoder(self.se)
            self.se = None
        if self.root:
            lib.free_huffman_tree(self.root)
            self.root = None


_table = None


def default_table():
    global _table
    if _table is None:
        _table = HuffmanTable()
    return _table


def use_table(path=TABLE_PATH):
    """Make `path` the table of the shuffman_* calls: the caller-owned objects the per-read
    functions receive, and the batch API's table (press_hip_load_table_file)."""
    global _table
    new = HuffmanTable(path)
    if _table is not None:
        _table.close()
    _table = new
    load_table(path)


def bound(method, n):
    """X_bound(n): what press/test.c mallocs for the compressed read."""
    lib = load_library()
    name = _SYMS[method][0]
    fn = getattr(lib, name)
    fn.restype = ctypes.c_uint64
    fn.argtypes = [ctypes.c_uint64 if _SYMS[method][3] in ("svb", "svb_nz") else ctypes.c_uint32]
    return int(fn(n))


def press(method, sig, cap=None):
    """X_press through the drop-in symbol. -> (ret, bytes); ret as the reference returns it
    (void methods: 0, or -1 when the library reports *nout = 0)."""
    lib = load_library()
    _, pname, _, kind = _SYMS[method]
    sig = np.ascontiguousarray(sig, dtype=np.int16)
    n = sig.size
    cap = bound(method, n) + 1024 if cap is None else int(cap)
    out = np.zeros(cap + 64, dtype=np.uint8)
    out[cap:] = 0xA5  # canary: nothing may be written past the capacity
    nout = ctypes.c_uint64(cap)
    fn = getattr(lib, pname)
    if kind in ("svb", "svb_nz", "vb"):
        fn.restype = None
        nt = ctypes.c_uint32 if (kind == "vb" or method == "svb12") else ctypes.c_uint64
        fn.argtypes = [ctypes.c_void_p, nt, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
        fn(sig.ctypes.data, n, out.ctypes.data, ctypes.byref(nout))
        ret = 0 if nout.value else -1
    elif kind == "int":
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
        ret = fn(sig.ctypes.data, n, out.ctypes.data, ctypes.byref(nout))
    else:
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p,
                       ctypes.POINTER(ctypes.c_uint64)]
        ret = fn(default_table().se, sig.ctypes.data, n, out.ctypes.data, ctypes.byref(nout))
    if not np.all(out[cap:] == 0xA5):
        raise PressError("%s wrote past its capacity" % pname)
    return ret, (out[: nout.value].tobytes() if ret == 0 else b"")


def depress(method, comp, n):
    """X_depress through the drop-in symbol. -> (ret, int16 array)"""
    lib = load_library()
    _, _, dname, kind = _SYMS[method]
    buf = np.frombuffer(bytes(comp) + b"\0" * 64, dtype=np.uint8).copy()
    out = np.zeros(n + 64, dtype=np.int16)
    out[n:] = 0x5A5A
    fn = getattr(lib, dname)
    ret = 0
    if kind == "svb_nz":
        fn.restype = None
        fn.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
        fn(buf.ctypes.data, n, out.ctypes.data)
        got = n
    elif kind == "svb":
        fn.restype = None
        nout = ctypes.c_uint64(n)
        fn.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
        fn(buf.ctypes.data, n, out.ctypes.data, ctypes.byref(nout))
        got = nout.value
    else:
        nout = ctypes.c_uint32(n)
        if kind == "shuff":
            fn.restype = ctypes.c_int
            fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p,
                           ctypes.POINTER(ctypes.c_uint32)]
            ret = fn(default_table().root, buf.ctypes.data, len(comp), out.ctypes.data, ctypes.byref(nout))
        else:
            fn.restype = None if kind == "vb" else ctypes.c_int
            fn.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32)]
            r = fn(buf.ctypes.data, len(comp), out.ctypes.data, ctypes.byref(nout))
            ret = 0 if kind == "vb" else r
        got = nout.value
        if kind == "vb" and got == 0:
            ret = -1
    if not np.all(out[n:] == 0x5A5A):
        raise PressError("%s wrote past the sample capacity" % dname)
    return ret, out[:got].copy()


# ---------------------------------------------------------------------------- batch API (torch CUDA tensors)

def use_torch_stream():
    """Make the batch calls enqueue on torch's current CUDA stream."""
    import torch

    lib = load_library()
    rc = lib.press_hip_set_device(torch.cuda.current_device())
    if rc:
        raise PressError(last_error())
    rc = lib.press_hip_set_stream(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc:
        raise PressError(last_error())


def load_table(path=TABLE_PATH):
    lib = load_library()
    if lib.press_hip_load_table_file(path.encode()):
        raise PressError(last_error())


def press_batch(method, sig, off, n, out, out_off, out_len):
    """Enqueue the compression of a batch.  All arguments are CUDA tensors: sig int16,
    off int64 (nreads read starts, multiples of 8), n int32 (nreads sample counts),
    out uint8, out_off int64 (nreads+1 slot bounds), out_len int64 (nreads; -1 = failed)."""
    lib = load_library()
    nreads = off.numel()
    rc = lib.press_hip_press_batch(_mid(method), sig.data_ptr(), off.data_ptr(), n.data_ptr(), nreads,
                                   sig.numel(), out.data_ptr(), out_off.data_ptr(), out_len.data_ptr(), 1)
    if rc:
        raise PressError(last_error())


def depress_batch(method, comp, in_off, in_len, sig, off, n, out_n):
    """Enqueue the decompression of a batch (CUDA tensors; out_n int32, -1 = failed)."""
    lib = load_library()
    nreads = off.numel()
    rc = lib.press_hip_depress_batch(_mid(method), comp.data_ptr(), in_off.data_ptr(), in_len.data_ptr(),
                                     nreads, sig.data_ptr(), off.data_ptr(), n.data_ptr(), sig.numel(),
                                     out_n.data_ptr(), 1)
    if rc:
        raise PressError(last_error())


def _layout(ns):
    """read starts padded to multiples of 8 samples -> (off uint64[nreads], total)"""
    ns = np.asarray(ns, dtype=np.int64)
    pad = (ns + 7) // 8 * 8
    off = np.zeros(ns.size, dtype=np.uint64)
    if ns.size > 1:
        off[1:] = np.cumsum(pad)[:-1]
    return off, int(pad.sum())


def press_batch_host(method, reads, caps=None):
    """Batch call with host buffers: reads = list of int16 arrays -> list of bytes / None."""
    lib = load_library()
    nreads = len(reads)
    ns = np.array([len(r) for r in reads], dtype=np.uint32)
    off, total = _layout(ns)
    sig = np.zeros(total + 64, dtype=np.int16)
    for r, o in zip(reads, off):
        sig[int(o): int(o) + len(r)] = r
    if caps is None:
        caps = [int(lib.press_hip_bound(_mid(method), int(x))) + 1024 for x in ns]
    out_off = np.zeros(nreads + 1, dtype=np.uint64)
    out_off[1:] = np.cumsum((np.asarray(caps, dtype=np.uint64) + 15) // 16 * 16)
    out = np.zeros(int(out_off[-1]) + 64, dtype=np.uint8)
    out_len = np.zeros(nreads, dtype=np.uint64)
    rc = lib.press_hip_press_batch(_mid(method), sig.ctypes.data, off.ctypes.data, ns.ctypes.data, nreads,
                                   total, out.ctypes.data, out_off.ctypes.data, out_len.ctypes.data, 0)
    if rc:
        raise PressError(last_error())
    return [None if int(l) == FAILED else out[int(o): int(o) + int(l)].tobytes()
            for o, l in zip(out_off[:-1], out_len)]


def depress_batch_host(method, streams, ns):
    """Batch decode with host buffers: streams = list of bytes, ns = sample counts / capacities."""
    lib = load_library()
    nreads = len(streams)
    in_len = np.array([len(s) for s in streams], dtype=np.uint64)
    in_off = np.zeros(nreads, dtype=np.uint64)
    if nreads > 1:
        in_off[1:] = np.cumsum(in_len)[:-1]
    comp = np.frombuffer(b"".join(streams) + b"\0" * 64, dtype=np.uint8).copy()
    ns = np.asarray(ns, dtype=np.uint32)
    off, total = _layout(ns)
    sig = np.zeros(total + 64, dtype=np.int16)
    out_n = np.zeros(nreads, dtype=np.uint32)
    rc = lib.press_hip_depress_batch(_mid(method), comp.ctypes.data, in_off.ctypes.data, in_len.ctypes.data,
                                     nreads, sig.ctypes.data, off.ctypes.data, ns.ctypes.data, total,
                                     out_n.ctypes.data, 0)
    if rc:
        raise PressError(last_error())
    return [None if int(k) == 0xFFFFFFFF else sig[int(o): int(o) + int(k)].copy()
            for o, k in zip(off, out_n)]


class HostBatch:
    """A batch in HOST memory laid out once, so that press() / depress() are nothing but the C calls
    (press_hip_press_batch / press_hip_depress_batch with device_resident = 0): what a C caller that
    keeps its reads in its own buffers pays - H2D, kernels, D2H - and what bench.py's PCIe-inclusive
    figure times.  pinned=True takes the buffers from press_hip_host_alloc (page-locked memory:
    the library then copies by DMA straight from / to them)."""

    def __init__(self, method, reads, pinned=False):
        lib = load_library()
        self.lib, self.mid, self.nreads = lib, _mid(method), len(reads)
        self.pinned = pinned
        self._held = []
        self.ns = np.array([len(r) for r in reads], dtype=np.uint32)
        self.off, self.total = _layout(self.ns)
        self.sig = self._alloc(self.total + 64, np.int16)
        self.sig[:] = 0
        for r, o in zip(reads, self.off):
            self.sig[int(o): int(o) + len(r)] = r
        caps = [int(lib.press_hip_bound(self.mid, int(x))) + 1024 for x in self.ns]
        self.out_off = np.zeros(self.nreads + 1, dtype=np.uint64)
        self.out_off[1:] = np.cumsum((np.asarray(caps, dtype=np.uint64) + 15) // 16 * 16)
        self.out = self._alloc(int(self.out_off[-1]) + 64, np.uint8)
        self.out_len = np.zeros(self.nreads, dtype=np.uint64)
        self.back = self._alloc(self.total + 64, np.int16)
        self.out_n = np.zeros(self.nreads, dtype=np.uint32)

    def _alloc(self, count, dtype):
        nbytes = int(count) * np.dtype(dtype).itemsize
        if not self.pinned:
            return np.zeros(count, dtype=dtype)
        self.lib.press_hip_host_alloc.restype = ctypes.c_void_p
        self.lib.press_hip_host_alloc.argtypes = [ctypes.c_uint64]
        p = self.lib.press_hip_host_alloc(nbytes)
        if not p:
            raise PressError(last_error())
        self._held.append(p)
        buf = (ctypes.c_uint8 * nbytes).from_address(p)
        return np.frombuffer(buf, dtype=dtype)

    def press(self):
        rc = self.lib.press_hip_press_batch(self.mid, self.sig.ctypes.data, self.off.ctypes.data,
                                            self.ns.ctypes.data, self.nreads, self.total, self.out.ctypes.data,
                                            self.out_off.ctypes.data, self.out_len.ctypes.data, 0)
        if rc:
            raise PressError(last_error())

    def streams(self):
        return [None if int(l) == FAILED else self.out[int(o): int(o) + int(l)].tobytes()
                for o, l in zip(self.out_off[:-1], self.out_len)]

    def depress(self):
        """decode the streams press() left in the slots of `out` into `back`"""
        in_off = np.ascontiguousarray(self.out_off[:-1])
        rc = self.lib.press_hip_depress_batch(self.mid, self.out.ctypes.data, in_off.ctypes.data,
                                              self.out_len.ctypes.data, self.nreads, self.back.ctypes.data,
                                              self.off.ctypes.data, self.ns.ctypes.data, self.total,
                                              self.out_n.ctypes.data, 0)
        if rc:
            raise PressError(last_error())

    def lossless(self):
        return bool(np.array_equal(self.out_n, self.ns)) and all(
            np.array_equal(self.back[int(o): int(o) + int(k)], self.sig[int(o): int(o) + int(k)])
            for o, k in zip(self.off, self.ns))

    def close(self):
        self.sig = self.out = self.back = None
        if self._held:
            self.lib.press_hip_host_free.argtypes = [ctypes.c_void_p]
            for p in self._held:
                self.lib.press_hip_host_free(p)
            self._held = []


def kernel_timing(enable=True):
    """(re)start / stop the HIP-event timing of the dominant kernel of each batch call"""
    lib = load_library()
    if lib.press_hip_kernel_timing(1 if enable else 0):
        raise PressError(last_error())


def kernel_times(which):
    """elapsed ms of the dominant kernel of every press (which=0) / depress (1) call so far"""
    lib = load_library()
    buf = (ctypes.c_float * 128)()
    lib.press_hip_kernel_times.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    n = lib.press_hip_kernel_times(which, buf, 128)
    return [buf[i] for i in range(n)]


# ---------------------------------------------------------------------------- BLOW5 input (host side)

class Blow5Reader:
    """press_hip_blow5_*: iterate over a BLOW5 file's signal fields as stored (svb-zd streams for
    signal method 1) - what depress_batch_host("slow5_svb_zd", ...) / the device batch API decode."""

    ID_LEN = 64

    def __init__(self, path):
        lib = load_library()
        lib.press_hip_blow5_last_error.restype = ctypes.c_char_p
        lib.press_hip_blow5_open.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p)]
        lib.press_hip_blow5_close.argtypes = [ctypes.c_void_p]
        lib.press_hip_blow5_methods.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
        lib.press_hip_blow5_next.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint64,
                                             ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.POINTER(ctypes.c_uint32)]
        self._h = ctypes.c_void_p()
        if lib.press_hip_blow5_open(path.encode(), ctypes.byref(self._h)):
            raise PressError(lib.press_hip_blow5_last_error().decode())
        rm, sm = ctypes.c_int(), ctypes.c_int()
        lib.press_hip_blow5_methods(self._h, ctypes.byref(rm), ctypes.byref(sm))
        self.record_method, self.signal_method = rm.value, sm.value

    def next_batch(self, max_reads=4096, arena_bytes=1 << 28):
        """-> list of (read_id, n_samples, signal field bytes); empty at the end of the file"""
        lib = load_library()
        arena = np.empty(arena_bytes, dtype=np.uint8)
        off = np.zeros(max_reads, dtype=np.uint64)
        ln = np.zeros(max_reads, dtype=np.uint64)
        ns = np.zeros(max_reads, dtype=np.uint32)
        ids = ctypes.create_string_buffer(max_reads * self.ID_LEN)
        got = ctypes.c_uint32()
        if lib.press_hip_blow5_next(self._h, max_reads, arena.ctypes.data, arena_bytes, off.ctypes.data,
                                    ln.ctypes.data, ns.ctypes.data, ids, ctypes.byref(got)):
            raise PressError(lib.press_hip_blow5_last_error().decode())
        out = []
        for k in range(got.value):
            rid = ids.raw[k * self.ID_LEN:(k + 1) * self.ID_LEN].split(b"\0")[0].decode()
            out.append((rid, int(ns[k]), arena[int(off[k]):int(off[k]) + int(ln[k])].tobytes()))
        return out

    def set_threads(self, n):
        """host threads that inflate records (0: as many as the host offers, at most 32)"""
        lib = load_library()
        lib.press_hip_blow5_threads.argtypes = [ctypes.c_void_p, ctypes.c_int]
        if lib.press_hip_blow5_threads(self._h, int(n)):
            raise PressError(lib.press_hip_blow5_last_error().decode())

    def next_arena(self, arena, off, ln, ns, max_reads):
        """the raw call: signal fields into the caller's arena (numpy uint8; page-locked memory from
        press.host_alloc for a fast copy to the device), offsets / lengths / sample counts into off / ln / ns
        (numpy uint64, uint64, uint32 of >= max_reads).  -> reads delivered (0 at the end of the file)"""
        lib = load_library()
        got = ctypes.c_uint32()
        if lib.press_hip_blow5_next(self._h, max_reads, arena.ctypes.data, arena.size, off.ctypes.data,
                                    ln.ctypes.data, ns.ctypes.data, None, ctypes.byref(got)):
            raise PressError(lib.press_hip_blow5_last_error().decode())
        return got.value

    def close(self):
        if self._h:
            load_library().press_hip_blow5_close(self._h)
            self._h = ctypes.c_void_p()


def _blow5_writer_api(lib):
    lib.press_hip_blow5_last_error.restype = ctypes.c_char_p
    lib.press_hip_blow5_create.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                           ctypes.POINTER(ctypes.c_void_p)]
    lib.press_hip_blow5_write.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p,
                                          ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64]
    lib.press_hip_blow5_write_batch.argtypes = [ctypes.c_void_p, ctypes.c_uint32] + [ctypes.c_void_p] * 6
    lib.press_hip_blow5_index.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.press_hip_blow5_finish.argtypes = [ctypes.c_void_p]


def _blow5_write_batch(lib, w, pres, sigs, posts):
    """press_hip_blow5_write_batch over lists of numpy uint8 arrays (posts may hold empty arrays)"""
    n = len(pres)
    PT = ctypes.c_void_p * n
    LT = ctypes.c_uint64 * n
    pre_p, sig_p, post_p = PT(), PT(), PT()
    pre_l, sig_l, post_l = LT(), LT(), LT()
    for k in range(n):
        pre_p[k], pre_l[k] = pres[k].ctypes.data, pres[k].size
        sig_p[k], sig_l[k] = (sigs[k].ctypes.data if sigs[k].size else None), sigs[k].size
        post_p[k], post_l[k] = (posts[k].ctypes.data if posts[k].size else None), posts[k].size
    if lib.press_hip_blow5_write_batch(w, n, pre_p, pre_l, sig_p, sig_l, post_p, post_l):
        raise PressError(lib.press_hip_blow5_last_error().decode())


def blow5_write_like(dst, like, fields, record_method=1, signal_method=1, index=False, ids=None):
    """A BLOW5 file with the header of `like` and one record per entry of `fields` (signal fields in the given
    signal method: svb-zd streams or int16 samples as bytes); every record takes the fixed fields and auxiliary
    fields of `like`'s first record and a read id of its own (ids, default "read-%08d" padded to the template's
    length).  For benchmarks and tests: a file of any size with realistic framing.  -> the read ids"""
    lib = load_library()
    _blow5_writer_api(lib)
    rd = Blow5Reader(like)
    lib.press_hip_blow5_next_records.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint64] + \
        [ctypes.c_void_p] * 5 + [ctypes.POINTER(ctypes.c_uint32)]
    cap = 1 << 26
    arena = np.empty(cap, dtype=np.uint8)
    ro, rl, sp, sl = (np.zeros(1, dtype=np.uint64) for _ in range(4))
    ns = np.zeros(1, dtype=np.uint32)
    got = ctypes.c_uint32()
    if lib.press_hip_blow5_next_records(rd._h, 1, arena.ctypes.data, cap, ro.ctypes.data, rl.ctypes.data, sp.ctypes.data,
                                        sl.ctypes.data, ns.ctypes.data, ctypes.byref(got)) or got.value != 1:
        rd.close()
        raise PressError("no template record in " + like)
    rec = arena[int(ro[0]):int(ro[0]) + int(rl[0])].copy()
    idl = int(rec[0]) | (int(rec[1]) << 8)
    fixed = rec[2 + idl:int(sp[0]) - 8].copy()          # read group, digitisation ... sampling rate
    post = rec[int(sp[0]) + int(sl[0]):].copy()         # auxiliary fields
    w = ctypes.c_void_p()
    if lib.press_hip_blow5_create(dst.encode(), rd._h, record_method, signal_method, ctypes.byref(w)):
        rd.close()
        raise PressError(lib.press_hip_blow5_last_error().decode())
    rd.close()
    out_ids = []
    try:
        if index:
            lib.press_hip_blow5_index(w, 1)
        B = 512
        for k0 in range(0, len(fields), B):
            pres, sigs, posts = [], [], []
            for k in range(k0, min(k0 + B, len(fields))):
                rid = (ids[k] if ids else "read-%08d" % k).encode()
                out_ids.append(rid.decode())
                pres.append(np.frombuffer(bytes([len(rid) & 255, len(rid) >> 8]) + rid + fixed.tobytes(), dtype=np.uint8))
                f = fields[k]
                sigs.append(f if isinstance(f, np.ndarray) else np.frombuffer(f, dtype=np.uint8))
                posts.append(post)
            _blow5_write_batch(lib, w, pres, sigs, posts)
    finally:
        if lib.press_hip_blow5_finish(w):
            raise PressError(lib.press_hip_blow5_last_error().decode())
    return out_ids


def blow5_transcode(src, dst, record_method=1, signal_method=1, codec=None, passthrough=False, index=False):
    """BLOW5 -> BLOW5 with the signal fields re-coded: codec(list of int16 arrays) -> list of svb-zd
    streams (default: the device, press_batch_host("slow5_svb_zd")); signal_method 0 writes the raw
    samples.  Signals of the source are decoded on the device when it stores them as svb-zd.
    passthrough: keep the signal fields as they are (source and target signal method must agree;
    no GPU involved - only the record framing / record compression changes).
    index: also write slow5lib's <dst>.idx (the reference's slow5_get then works on the transcoded file).
    Returns the number of reads written."""
    lib = load_library()
    _blow5_writer_api(lib)
    rd = Blow5Reader(src)
    lib.press_hip_blow5_next_records.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint64] + \
        [ctypes.c_void_p] * 5 + [ctypes.POINTER(ctypes.c_uint32)]
    lib.press_hip_blow5_create.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                           ctypes.POINTER(ctypes.c_void_p)]
    lib.press_hip_blow5_write.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p,
                                          ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64]
    lib.press_hip_blow5_finish.argtypes = [ctypes.c_void_p]
    w = ctypes.c_void_p()
    if lib.press_hip_blow5_create(dst.encode(), rd._h, record_method, signal_method, ctypes.byref(w)):
        rd.close()
        raise PressError(lib.press_hip_blow5_last_error().decode())
    total = 0
    try:
        if index:
            lib.press_hip_blow5_index(w, 1)
        maxr, cap = 1024, 1 << 28
        arena = np.empty(cap, dtype=np.uint8)
        ro, rl, sp, sl = (np.zeros(maxr, dtype=np.uint64) for _ in range(4))
        ns = np.zeros(maxr, dtype=np.uint32)
        got = ctypes.c_uint32()
        while True:
            if lib.press_hip_blow5_next_records(rd._h, maxr, arena.ctypes.data, cap, ro.ctypes.data, rl.ctypes.data,
                                                sp.ctypes.data, sl.ctypes.data, ns.ctypes.data, ctypes.byref(got)):
                raise PressError(lib.press_hip_blow5_last_error().decode())
            if got.value == 0:
                break
            recs = [arena[int(ro[k]):int(ro[k]) + int(rl[k])] for k in range(got.value)]
            fields = [r[int(sp[k]):int(sp[k]) + int(sl[k])].tobytes() for k, r in enumerate(recs)]
            if passthrough:
                if rd.signal_method != signal_method:
                    raise PressError("passthrough needs the same signal method on both sides")
                out = fields
                sigs = None
            elif rd.signal_method == 1:
                sigs = depress_batch_host("slow5_svb_zd", fields, [int(x) for x in ns[:got.value]])
                if any(s is None for s in sigs):
                    raise PressError("a signal of %s does not decode" % src)
            else:
                sigs = [np.frombuffer(f, dtype=np.int16) for f in fields]
            if passthrough:
                pass
            elif signal_method == 1:
                out = (codec or (lambda reads: press_batch_host("slow5_svb_zd", reads)))(sigs)
            else:
                out = [np.ascontiguousarray(s, dtype=np.int16).tobytes() for s in sigs]
            _blow5_write_batch(lib, w, [r[:int(sp[k]) - 8] for k, r in enumerate(recs)],
                               [np.frombuffer(out[k], dtype=np.uint8) for k in range(got.value)],
                               [r[int(sp[k]) + int(sl[k]):] for k, r in enumerate(recs)])
            total += got.value
    finally:
        rd.close()
        if lib.press_hip_blow5_finish(w):
            raise PressError(lib.press_hip_blow5_last_error().decode())
    return total

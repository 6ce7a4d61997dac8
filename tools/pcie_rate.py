"""PCIe-inclusive rate of the batch API with HOST buffers (device_resident = 0): one call =
H2D of the samples + kernels + D2H of the streams (and the reverse).  Never bench.py's `value`."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (brings up the HIP runtime the library links against)
from honours_amd import press, synth

m = sys.argv[1] if len(sys.argv) > 1 else "svb12_zd"
nreads = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
lib = press.load_library()
if m.startswith("shuffman"):
    press.load_table()
sig, off = synth.synth_batch(1, 0, nreads)
reads = [sig[int(off[k]):int(off[k + 1])] for k in range(nreads)]
ns = np.array([len(r) for r in reads], dtype=np.uint32)
o, total = press._layout(ns)
buf = np.zeros(total + 64, dtype=np.int16)
for r, x in zip(reads, o):
    buf[int(x):int(x) + len(r)] = r
caps = np.array([int(lib.press_hip_bound(press.METHODS[m], int(x))) + 1024 for x in ns], dtype=np.uint64)
out_off = np.zeros(nreads + 1, dtype=np.uint64)
out_off[1:] = np.cumsum((caps + 15) // 16 * 16)
out = np.zeros(int(out_off[-1]) + 64, dtype=np.uint8)
out_len = np.zeros(nreads, dtype=np.uint64)
back = np.zeros_like(buf)
out_n = np.zeros(nreads, dtype=np.uint32)
raw = 2 * int(ns.sum())
for it in range(3):
    t0 = time.perf_counter()
    assert lib.press_hip_press_batch(press.METHODS[m], buf.ctypes.data, o.ctypes.data, ns.ctypes.data, nreads, total,
                                     out.ctypes.data, out_off.ctypes.data, out_len.ctypes.data, 0) == 0
    t1 = time.perf_counter()
    in_off = out_off[:-1].copy()
    assert lib.press_hip_depress_batch(press.METHODS[m], out.ctypes.data, in_off.ctypes.data, out_len.ctypes.data, nreads,
                                       back.ctypes.data, o.ctypes.data, ns.ctypes.data, total, out_n.ctypes.data, 0) == 0
    t2 = time.perf_counter()
assert np.array_equal(back[:total], buf[:total])
print("%s %d reads %.1f MB raw: press %.1f MB/s, depress %.1f MB/s (host buffers, pageable, incl. PCIe)" %
      (m, nreads, raw / 1e6, raw / (t1 - t0) / 1e6, raw / (t2 - t1) / 1e6))

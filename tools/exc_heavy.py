"""how the exception-split path behaves when MANY samples are exceptions (not NA12878-like):
batch of 512 reads x 100k samples with the given exception rate; time of press + depress (host batch API,
so PCIe is included: compare the rates against each other only)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from honours_amd import press

rate = float(sys.argv[1]) if len(sys.argv) > 1 else 0.05
m = sys.argv[2] if len(sys.argv) > 2 else "vbe21_zd"
rng = np.random.default_rng(1)
reads = []
for r in range(512):
    d = rng.integers(-40, 41, size=100000)
    ex = rng.random(100000) < rate
    d[ex] = rng.integers(-3000, 3000, size=int(ex.sum()))
    reads.append((np.cumsum(d) % 4000).astype(np.int16))
press.load_library()
if m.startswith("shuffman"):
    press.load_table()
caps = [len(r) * 4 + 4096 for r in reads]
for it in range(2):
    t0 = time.perf_counter()
    st = press.press_batch_host(m, reads, caps=caps)
    t1 = time.perf_counter()
    back = press.depress_batch_host(m, st, [len(r) for r in reads])
    t2 = time.perf_counter()
assert all(b is not None and np.array_equal(b, r) for b, r in zip(back, reads))
print("%s exception rate %.3f: press %.1f ms, depress %.1f ms for %d MB raw" % (m, rate, (t1 - t0) * 1e3, (t2 - t1) * 1e3, sum(len(r) for r in reads) * 2 // 1000000))

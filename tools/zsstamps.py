"""diagnostic: where the waves of k_zs_hdecode spend their time (HUF_STAMPS build: tools/build_variants.sh
"hstamp:-DHUF_STAMPS"); s_memtime ticks of a wave summed per phase"""
import ctypes, os, sys, runpy
import numpy as np
here = os.path.dirname(os.path.abspath(__file__))
os.environ["PRESS_HIP_LIB"] = os.path.join(here, "bin", "libpress_%s.so" % (sys.argv[1] if len(sys.argv) > 1 else "hstamp"))
sys.path.insert(0, os.path.dirname(here))
sys.argv = ["bench.py", "--method", "zstd_svb_zd", "--steps", "3", "--warmup", "1", "--no-cpu", "--no-sub"]
buf = np.zeros(8, dtype=np.uint64)
try:
    runpy.run_path(os.path.join(os.path.dirname(here), "bench.py"), run_name="__main__")
except SystemExit:
    pass
from honours_amd import press  # (after bench.py: torch initialises the GPU before the library is loaded)
lib = press.load_library()
lib.press_hip_zs_stamps.argtypes = [ctypes.c_void_p]
assert lib.press_hip_zs_stamps(buf.ctypes.data) == 0
names = ["tables", "stream headers", "staging (loads -> LDS)", "decode loop", "stores"]
tot = float(sum(int(x) for x in buf))
for i, n in enumerate(names):
    print("%-26s %14d ticks  %5.1f %%" % (n, int(buf[i]), 100.0 * int(buf[i]) / tot))

# k_zs_walk: ticks of each read's wave against the read's length
wt = np.zeros(8192, dtype=np.uint32)
lib.press_hip_zs_walk_ticks.argtypes = [ctypes.c_void_p]
assert lib.press_hip_zs_walk_ticks(wt.ctypes.data) == 0
from honours_amd import synth
n, _ = synth.read_lengths(20261004, 0, 8192)
blocks = (n + 16383) // 16384
print("walk: ticks per read  min %d  median %d  p90 %d  max %d" % (wt.min(), np.median(wt), np.percentile(wt, 90), wt.max()))
A = np.vstack([np.ones(8192), blocks]).T
coef = np.linalg.lstsq(A, wt.astype(np.float64), rcond=None)[0]
print("walk: ticks ~ %.0f + %.0f per data block (least squares); longest read: %d blocks, %d ticks" % (coef[0], coef[1], blocks.max(), wt[np.argmax(blocks)]))
for lo, hi in ((1, 2), (3, 5), (6, 10), (11, 20), (21, 40), (41, 200)):
    m = (blocks >= lo) & (blocks <= hi)
    print("  blocks %3d..%3d: reads %5d  mean ticks %9.0f" % (lo, hi, m.sum(), wt[m].mean() if m.any() else 0))

ws = np.zeros(8, dtype=np.uint64)
lib.press_hip_zs_walk_stamps.argtypes = [ctypes.c_void_p]
assert lib.press_hip_zs_walk_stamps(ws.ctypes.data) == 0
wn = ["window fetches", "raw / RLE pieces queued", "tree description fetched", "tree description read", "tree stored",
      "Huffman blocks queued", "end of the walk", "the walk's own code"]
tot = float(ws.sum())
for i, nme in enumerate(wn):
    print("walk %-28s %14d ticks  %5.1f %%" % (nme, int(ws[i]), 100.0 * int(ws[i]) / tot))

ts = np.zeros(8, dtype=np.uint64)
lib.press_hip_zs_table_stamps.argtypes = [ctypes.c_void_p]
assert lib.press_hip_zs_table_stamps(ts.ctypes.data) == 0
tn = ["key ranks, counts, order", "tree (two queues)", "depths", "length limit", "lengths, codes", "description (FSE)", "rest (direct description)"]
tot = float(ts.sum())
for i, nme in enumerate(tn):
    print("table %-28s %14d ticks  %5.1f %%" % (nme, int(ts[i]), 100.0 * int(ts[i]) / tot))

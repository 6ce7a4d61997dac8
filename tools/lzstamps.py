"""diagnostic: where the full walk (k_zs_walk<false>) spends its time on ZSTD_compress's own frames (HUF_STAMPS build)"""
import ctypes, os, sys, runpy
import numpy as np
here = os.path.dirname(os.path.abspath(__file__))
os.environ["PRESS_HIP_LIB"] = os.path.join(here, "bin", "libpress_hstamp.so")
sys.path.insert(0, os.path.dirname(here))
sys.argv = ["lzframes.py", "2048", "3"]
try:
    runpy.run_path(os.path.join(here, "lzframes.py"), run_name="__main__")
except SystemExit:
    pass
from honours_amd import press
lib = press.load_library()
ws = np.zeros(8, dtype=np.uint64)
lib.press_hip_zs_walk_stamps.argtypes = [ctypes.c_void_p]
assert lib.press_hip_zs_walk_stamps(ws.ctypes.data) == 0
wn = ["window fetches", "copies / fills; the three FSE tables of a block with sequences", "tree description fetched", "tree description read",
      "tree stored", "Huffman blocks queued", "sequences read (+ end of the walk)", "the walk's own code"]
tot = float(ws.sum())
for i, nme in enumerate(wn):
    print("walk %-64s %14d ticks  %5.1f %%" % (nme, int(ws[i]), 100.0 * int(ws[i]) / tot))

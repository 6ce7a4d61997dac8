"""diagnostic: where the waves of k_huf_sync / k_huf_emit spend their time (HUF_STAMPS build:
tools/build_variants.sh "hstamp:-DHUF_STAMPS"); s_memtime ticks of lane 0 summed per phase over all units"""
import ctypes, os, sys, runpy
import numpy as np
here = os.path.dirname(os.path.abspath(__file__))
os.environ["PRESS_HIP_LIB"] = os.path.join(here, "bin", "libpress_%s.so" % (sys.argv[1] if len(sys.argv) > 1 else "hstamp"))
sys.path.insert(0, os.path.dirname(here))
sys.argv = ["bench.py", "--steps", "3", "--warmup", "1", "--no-cpu", "--no-sub"]
buf = np.zeros(32, dtype=np.uint64)
import io, contextlib
try:
    runpy.run_path(os.path.join(os.path.dirname(here), "bench.py"), run_name="__main__")
except SystemExit:
    pass
from honours_amd import press  # (after bench.py: torch initialises the GPU before the library is loaded)
lib = press.load_library()
lib.press_hip_huf_stamps.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
assert lib.press_hip_huf_stamps(buf.ctypes.data, 32) == 0
names = {0: "sync ticket", 1: "sync loads", 2: "sync run-up", 3: "sync own pass", 4: "sync list", 5: "sync scans + record stores",
         8: "emit tickets + next units asked for", 9: "emit next units land (after decode)", 10: "emit scan", 11: "emit decode", 12: "emit plan", 13: "emit samples"}
names[14] = "emit rest (waits behind the stores)"
for grp in ((0, 1, 2, 3, 5, 4), (8, 9, 10, 11, 12, 13, 14)):
    tot = float(sum(int(buf[i]) for i in grp))
    for i in grp:
        print("%-22s %14d ticks  %5.1f %%" % (names[i], int(buf[i]), 100.0 * int(buf[i]) / tot))
    print()

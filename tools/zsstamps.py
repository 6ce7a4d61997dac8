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

"""diagnostic: per-phase wave-0 timing of k_svb_decode_chunked / k_low_decode_chunked (DEC_STAMPS build);
extra arguments go to bench.py (e.g. --method vbe21_zd)"""
import ctypes, os, sys, subprocess
import numpy as np
os.environ["PRESS_HIP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "libpress_stamps.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py", "--steps", "1", "--warmup", "1", "--no-cpu"] + sys.argv[1:]
import runpy
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
from honours_amd import press
lib = press.load_library()
n = 28000
buf = np.zeros(n * 8, dtype=np.uint64)
lib.press_hip_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
assert lib.press_hip_debug_stamps(buf.ctypes.data, n * 8) == 0
s = buf.reshape(n, 8).astype(np.int64)
ok = (s[:, 6] > 0) & (s[:, 0] > 0)
s = s[ok]
names = ["start->ticket", "ticket->desc", "desc->data+sum", "sum->barrier", "barrier->lookback", "lookback->stores issued"]
d = np.diff(s[:, :7], axis=1)
print("chunks", len(s), "clock ticks (s_memtime, 100 MHz?)")
for i, nm in enumerate(names):
    print("%-26s mean %9.1f  p50 %9.1f  p90 %9.1f" % (nm, d[:, i].mean(), np.percentile(d[:, i], 50), np.percentile(d[:, i], 90)))
tot = s[:, 6] - s[:, 0]
print("total mean", tot.mean(), "kernel span", s[:, 6].max() - s[:, 0].min())

import ctypes, os, sys
os.environ["PRESS_HIP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "libpress_hufdbg.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py", "--steps", "1", "--warmup", "1", "--no-cpu", "--method", "shuffman_vbe21_zd", "--reads", "1024"]
import runpy
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
from honours_amd import press
lib = press.load_library()
buf = (ctypes.c_ulonglong * 8)()
lib.press_hip_debug_huff(buf)
print("tiles", buf[0], "rounds", buf[1], "redecodes", buf[2], "rounds/tile", buf[1] / max(1, buf[0]), "redecodes/tile", buf[2] / max(1, buf[0]))

"""diagnostic: per-phase clock totals of k_huff_decode_tiles (HUF_DEBUG build, tools/build_variants.sh hufdbg:-DHUF_DEBUG)"""
import ctypes, os, sys
os.environ["PRESS_HIP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "libpress_hufdbg.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py", "--steps", "1", "--warmup", "1", "--no-cpu", "--no-check", "--method", "shuffman_vbe21_zd", "--reads", "8192"] + sys.argv[1:]
import runpy
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
from honours_amd import press  # after bench.py has brought up torch's HIP runtime
lib = press.load_library()
buf = (ctypes.c_ulonglong * 16)()
lib.press_hip_debug_huff(buf)
names = ["ticket", "stage", "pass0", "roundsA", "hint wait", "roundsB", "prefix+final wait", "output"]
tiles = max(1, buf[8])
tot = sum(buf[:8])
print("tiles", buf[8], "rounds/tile", buf[9] / tiles, "redecodes/tile", buf[11] / tiles, "wrong hints", buf[10])
for i, nm in enumerate(names):
    print("%-20s %10.0f ticks/tile  %5.1f %%" % (nm, buf[i] / tiles, 100.0 * buf[i] / max(1, tot)))
print("total ticks/tile", tot / tiles)
print("look-back polls that had to be repeated: %.3f per tile" % (buf[14] / tiles))
print("inside the rounds: detection %.0f ticks/tile, re-decode %.0f ticks/tile" % (buf[12] / tiles, buf[13] / tiles))

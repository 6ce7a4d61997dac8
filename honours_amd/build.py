"""Build libpress_hip.so (HIP kernels + C ABI) for gfx950, in-tree.

    python -m honours_amd.build

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpress_hip.so")
SOURCES = ["press_sections.hip", "press_chunked.hip", "press_huffman.hip", "press_rc.hip", "press_zstd.hip", "press_abi.hip", "blow5_reader.cpp"]
HEADERS = ["press_internal.h", "press_packed.h", "zs_table.h", os.path.join("..", "..", "include", "press_hip.h")]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False):
    """Compile the library if it is missing or older than its sources. Returns its path."""
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall",
           "-Wno-unused-function", "-Wno-cast-align"] + [os.path.join(CSRC, f) for f in SOURCES] + ["-o", LIB, "-ldl"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

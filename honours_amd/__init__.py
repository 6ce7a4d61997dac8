"""honours_amd - MI355X (gfx950) implementation of the per-read compression hot path of
sashajenner/honours `press/` (zig-zag delta -> StreamVByte / exception split -> static
Huffman), behind the reference's own C interface.

The product is the shared library ``libpress_hip.so`` (HIP kernels + C ABI declared in
``include/press_hip.h``); this package is the thin host-side mirror of that interface
for Python callers (tests, bench.py).  There is no CPU implementation here: importing
``honours_amd.press`` without the built library raises.
"""
__all__ = ["press", "synth", "build"]

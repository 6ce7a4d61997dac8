#!/usr/bin/env python3
"""the device's zstd reader on libzstd's own frames (bench.py's libzstd_frames leg alone; for rocprofv3):
    python3 tools/lzframes.py [reads] [steps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from honours_amd import press, synth  # noqa: E402

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
assert torch.cuda.is_available()
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
press.load_library()
press.use_torch_stream()
b = bench.Batch(torch, press, synth, 20261004, 0, reads, dev, None)
press.kernel_timing(True)
print(bench.libzstd_frames(torch, press, b, steps, nreads=reads))
kt = press.kernel_times(1)
print('k_zs_hdecode ms per call:', [round(x, 3) for x in kt])
press.kernel_timing(False)
del b
torch.cuda.synchronize()
press.load_library().press_hip_shutdown()

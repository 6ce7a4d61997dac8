#!/usr/bin/env python3
"""bench.py's e2e_blow5 leg alone:  python3 tools/e2e_blow5.py [reads] [reads per batch]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from honours_amd import press, synth  # noqa: E402

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
per = int(sys.argv[2]) if len(sys.argv) > 2 else 512
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
press.load_library()
press.use_torch_stream()
b = bench.Batch(torch, press, synth, 20261004, 0, reads, dev, None)
print(json.dumps(bench.e2e_blow5(torch, press, b, "shuffman_vbe21_zd", nreads=reads, batch_reads=per)))

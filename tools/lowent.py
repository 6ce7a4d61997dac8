#!/usr/bin/env python3
"""How does the Huffman decode behave on LOW-entropy stretches (a nanopore read's "stall": small deltas, 4-bit codes,
more than EMIT_STG symbols per 64 subsequences)?  Reads whose first `frac` of samples have |delta| <= 1, the rest
NA12878-like; device-resident press + depress times.   python3 tools/lowent.py [frac ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from honours_amd import press, synth

mode = "low"
args = sys.argv[1:]
if args and args[0] in ("low", "high"):
    mode, args = args[0], args[1:]
fracs = [float(x) for x in args] or [0.0, 0.01, 0.1, 1.0]
span = 1 if mode == "low" else 120  # high: deltas uniform in [-120, 120] - the table's long codes (13 .. 22 bits) all the time
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
press.load_library(); press.use_torch_stream(); press.load_table()
R = 2048
b = bench.Batch(torch, press, synth, 20261004, 0, R, dev, None)
base = b.sig.clone()
g = torch.Generator(device=dev); g.manual_seed(5)
for frac in fracs:
    sig = base.clone()
    if frac > 0:
        # overwrite the first part of every read by a small-step walk
        for r in range(R):
            n0 = int(b.starts[r]); k = int(int(b.n[r]) * frac)
            if k > 1:
                steps = torch.randint(-span, span + 1, (k,), device=dev, generator=g, dtype=torch.int16)
                steps[0] = 500
                sig[n0:n0 + k] = (torch.cumsum(steps.to(torch.int32), 0) % 3000).to(torch.int16) if span > 1 else torch.cumsum(steps.to(torch.int32), 0).to(torch.int16)
    b.sig = sig
    caps, d_out, d_out_off, d_in_off = b.arena(torch, press, "shuffman_vbe21_zd")
    if mode == "high":  # (the reference's bound is too small for such reads: give the slots room)
        caps = np.array([4 * int(x) + 4096 for x in b.n], dtype=np.int64) // 128 * 128
        oo = np.concatenate([[0], np.cumsum(caps)])
        d_out = torch.empty(int(oo[-1]) + 64, dtype=torch.uint8, device=dev)
        d_out_off = torch.from_numpy(oo).to(dev); d_in_off = d_out_off[:-1].contiguous()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    for it in range(3):
        if it == 2: ev[0].record()
        press.press_batch("shuffman_vbe21_zd", b.sig, b.d_off, b.d_n, d_out, d_out_off, b.d_len)
        if it == 2: ev[1].record()
        press.depress_batch("shuffman_vbe21_zd", d_out, d_in_off, b.d_len, b.d_back, b.d_off, b.d_n, b.d_outn)
    ev[2].record(); torch.cuda.synchronize()
    assert torch.equal(b.d_back, b.sig)
    comp = int(b.d_len.sum())
    print(mode + "-entropy fraction %.2f: ratio %.3f press %.3f ms depress %.3f ms (%d reads, %d MB raw)" % (
        frac, b.raw_bytes / comp, ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2]), R, b.raw_bytes // 1000000))

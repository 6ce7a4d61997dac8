#!/usr/bin/env python3
"""Stress of the static-Huffman batch path on the GPU (gpurun -- python3 tools/stress_huff.py [ITER] [READS] [METHOD,METHOD]):
different synthetic batches (seed, read count, fixed or natural lengths), each compressed and decompressed
REPEAT times on the device; every decode must give back the samples and every encode the same bytes as the
first one of its batch (the kernels hand out work by tickets and atomics: no run may depend on their order).
Prints one line per batch; exits non-zero at the first difference."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def stress(iters, reads, repeat=6, verbose=True, methods=("shuffman_vbe21_zd", "shuffman_vbsse21_zd")):
    """-> number of batches run; raises AssertionError at the first difference"""
    import torch

    import bench
    from honours_amd import press, synth

    dev = torch.device("cuda:0")
    press.load_library()
    press.load_table()
    rng = np.random.default_rng(7)
    for it in range(iters):
        seed = int(rng.integers(1, 1 << 30))
        R = int(rng.integers(1, reads + 1))
        fixed = None if it % 3 else int(rng.integers(1, 300000))
        b = bench.Batch(torch, press, synth, seed, int(rng.integers(0, 1 << 20)), R, dev, fixed)
        for m in methods:
            caps, d_out, d_out_off, d_in_off = b.arena(torch, press, m)
            first = None
            for k in range(repeat):
                d_out.zero_()
                press.press_batch(m, b.sig, b.d_off, b.d_n, d_out, d_out_off, b.d_len)
                b.d_back.fill_(-1)
                press.depress_batch(m, d_out, d_in_off, b.d_len, b.d_back, b.d_off, b.d_n, b.d_outn)
                torch.cuda.synchronize()
                lens = b.d_len.cpu().numpy().copy()
                ok = bool((lens > 0).all()) and bool((b.d_outn.cpu().numpy() == b.n.astype(np.int32)).all())
                # samples of every read (the gaps between the reads' slots are not the decoder's to write)
                starts = b.starts[:-1].astype(np.int64)
                back = b.d_back.cpu().numpy()
                sig = b.sig.cpu().numpy()
                for r in range(R):
                    if not np.array_equal(back[starts[r]: starts[r] + b.n[r]], sig[starts[r]: starts[r] + b.n[r]]):
                        ok = False
                        print("read", r, "n", int(b.n[r]), "differs")
                        break
                out = d_out.cpu().numpy()
                sig_out = (lens.tobytes(), out.tobytes())
                if first is None:
                    first = sig_out
                elif first != sig_out:
                    ok = False
                    print("encode run", k, "differs from run 0")
                assert ok, ("seed", seed, "reads", R, "fixed", fixed, m, "run", k)
        if verbose:
            print("batch %3d seed %10d reads %5d fixed %s samples %d ok" % (it, seed, R, fixed, b.total_samples), flush=True)
    return iters


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    reads = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    methods = tuple(sys.argv[3].split(",")) if len(sys.argv) > 3 else ("shuffman_vbe21_zd", "shuffman_vbsse21_zd")
    print("all", stress(iters, reads, methods=methods), "batches ok", methods)


if __name__ == "__main__":
    main()

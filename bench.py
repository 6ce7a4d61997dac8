#!/usr/bin/env python3
"""bench.py - the reference's headline measurement on MI355X.

Metric (BASELINE.json): raw-signal MB/s for compress + decompress (MB = 1e6 bytes of
int16 signal, press/test.c's timing of X_press + X_depress) and the compression ratio,
on NA12878-like reads.  The NA12878 500k-read set is not available offline, so the
workload is its synthetic stand-in (honours_amd/synth.py: lengths, first samples,
zig-zag-delta symbol statistics and exception rate of the published tables).

One "step" = one pass of the hot path over one device-resident batch of reads:
press_batch then depress_batch.  `value` = raw signal bytes of the batch / step time,
whole job over all ranks (each rank owns its own reads: weak scaling, no data-path
collective; one RCCL all-reduce of the {raw, compressed, reads} totals at the end).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--method svb12_zd] [--reads R]

N > 1 is launched by the driver with torch.distributed.run (one rank per GPU).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

VBZ_RATIO = 2.928430  # data/reads.blow5.test:11 (zstd-svb-zd on NA12878)
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)

WORKLOADS = {
    "svb12_zd": "NA12878-like synthetic reads, zig-zag-delta + svb16 pack only (svb12_zd, no entropy stage)",
    "svb_zd": "NA12878-like synthetic reads, zig-zag-delta + svb32 pack (svb_zd)",
    "vbe21_zd": "NA12878-like synthetic reads, exception split (vbe21_zd)",
    "hasgam_vbsse21_zdq": "NA12878-like synthetic reads, ex-zd (hasgam_vbsse21_zdq)",
    "shuffman_vbe21_zd": "NA12878-like synthetic reads, ex split + static NA12878_zd Huffman (shuffman_vbe21_zd)",
    "slow5_svb_zd": "NA12878-like synthetic reads, BLOW5's signal codec (slow5lib svb-zd: u32 count + svb32 of zig-zag deltas)",
    "zstd_svb_zd": "NA12878-like synthetic reads, full VBZ pipeline zstd(svb-zd) with the zstd frames made and read on the device (config 3)",
    "zstd_svb12_zd": "NA12878-like synthetic reads, zstd(svb16-zd) with the zstd frames made and read on the device",
    "zstd_hasgam_vbsse21_zdq": "NA12878-like synthetic reads, zstd(ex-zd) with the zstd frames made and read on the device",
    "rc_vbe21_zd": "NA12878-like synthetic reads, ex split + order-0 adaptive range coder (rc_vbe21_zd; serial per read by format)",
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--method", default="svb12_zd", choices=sorted(WORKLOADS))
    ap.add_argument("--reads", type=int, default=8192, help="reads per GPU per step")
    ap.add_argument("--fixed-len", type=int, default=None, help="config 5: fixed read length")
    ap.add_argument("--seed", type=int, default=20261004)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-check", action="store_true", help=argparse.SUPPRESS)  # diagnostic kernel builds only
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from honours_amd import press, shard, synth

    rank, world, local_rank = shard.world_info()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    press.load_library()
    press.use_torch_stream()
    if args.method.startswith("shuffman"):
        press.load_table()
    m = args.method

    # ---- synthetic batch, generated on the device (never staged through PCIe)
    first_read, R = shard.weak_shard(args.reads, rank)
    sig, starts, n = synth.synth_batch_torch(args.seed, first_read, R, dev, fixed_len=args.fixed_len, align=64)
    sig = torch.cat([sig, torch.zeros(64, dtype=torch.int16, device=dev)])
    total_samples = int(n.sum())
    raw_bytes = 2 * total_samples
    d_off = torch.from_numpy(starts[:-1].astype(np.int64)).to(dev)
    d_n = torch.from_numpy(n.astype(np.int32)).to(dev)
    caps = np.array([press.bound(m, int(x)) for x in n], dtype=np.int64)
    caps = (caps + 64 + 127) // 128 * 128
    out_off = np.concatenate([[0], np.cumsum(caps)])
    d_out = torch.empty(int(out_off[-1]) + 64, dtype=torch.uint8, device=dev)
    d_out_off = torch.from_numpy(out_off).to(dev)
    d_in_off = d_out_off[:-1].contiguous()
    d_len = torch.zeros(R, dtype=torch.int64, device=dev)
    d_back = torch.zeros_like(sig)
    d_outn = torch.zeros(R, dtype=torch.int32, device=dev)

    def step(ev=None):
        if ev is not None:
            ev[0].record()
        press.press_batch(m, sig, d_off, d_n, d_out, d_out_off, d_len)
        if ev is not None:
            ev[1].record()
        press.depress_batch(m, d_out, d_in_off, d_len, d_back, d_off, d_n, d_outn)
        if ev is not None:
            ev[2].record()

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    # correctness of what is being timed: lossless on the device, sizes sane
    lens = d_len.cpu().numpy()
    if not args.no_check:
        assert (lens > 0).all() and (lens < caps).all(), "a read failed to compress"
        assert bool((d_outn.cpu() == torch.from_numpy(n.astype(np.int32))).all()), "sample counts"
        assert torch.equal(d_back, sig), "round trip is not lossless"
    comp_bytes = int(lens.sum())

    events = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    press.kernel_timing(True)  # HIP events around the dominant kernel of each call, on its launch stream
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(events[k])
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()

    elapsed = t1 - t0
    press_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in events]))
    depress_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in events]))
    kpress = press.kernel_times(0)
    kdepress = press.kernel_times(1)
    press.kernel_timing(False)
    kpress_ms = float(np.mean(kpress)) if kpress else press_ms
    kdepress_ms = float(np.mean(kdepress)) if kdepress else depress_ms
    # the only collective: 24 bytes of totals (+ the slowest rank's time) over RCCL
    raw_all, comp_all, reads_all, elapsed = shard.reduce_totals(raw_bytes, comp_bytes, R, elapsed, dev)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = raw_all / (elapsed / args.steps) / 1e6
        ratio = raw_all / comp_all
        # Roofline: algorithmic bytes per launch = sum over the batch of
        # 2n (int16 samples) + c (compressed stream), read + written once; the dominant kernel is
        # the longer of the two main kernels, timed with HIP events on its own launch stream.
        alg = raw_bytes + comp_bytes
        kern = {"svb12_zd": ("k_svb_encode_chunked<false,true>", "k_svb_decode_chunked<false,true>"),
                "svb_zd": ("k_svb_encode_chunked<true,true>", "k_svb_decode_chunked<true,true>"),
                "slow5_svb_zd": ("k_svb_encode_chunked<true,true,true>", "k_svb_decode_chunked<true,true,true>"),
                "rc_vbe21_zd": ("k_rcs_encode", "k_rcs_decode"),
                "zstd_svb_zd": ("k_zs_encode", "k_zs_hdecode"),
                "zstd_svb12_zd": ("k_zs_encode", "k_zs_hdecode"),
                "zstd_hasgam_vbsse21_zdq": ("k_zs_encode", "k_zs_hdecode"),
                "shuffman_vbe21_zd": ("k_huff_encode_chunked", "k_huff_decode_tiles")}.get(
                    m, ("k_low_encode_chunked", "k_low_decode_chunked<false>"))
        traffic = measured_traffic(m, R, args.seed, args.fixed_len)

        def roof(name, ms, call_ms, key):
            gbps = alg / (ms * 1e-3) / 1e9
            return {"kernel": name, "bound": "hbm", "achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS,
                    "unit": "GB/s", "frac": round(gbps / HBM_PEAK_GBPS, 4),
                    "traffic": traffic.get(key) if traffic else None,
                    "algorithmic_bytes_per_launch": alg, "avg_launch_ms": round(ms, 4),
                    "whole_call_ms": round(call_ms, 4)}

        r_press = roof(kern[0], kpress_ms, press_ms, "press")
        r_depress = roof(kern[1], kdepress_ms, depress_ms, "depress")
        dominant, other = (r_depress, r_press) if kdepress_ms >= kpress_ms else (r_press, r_depress)
        out = {
            "metric": "raw-signal MB/s (compress+decompress)",
            "value": round(value, 1),
            "unit": "MB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u16",
            "data": "synthetic",
            "config": {
                "workload": WORKLOADS[m],
                "method": m,
                "reads_per_gpu": R,
                "samples_per_gpu": total_samples,
                "mean_read_len": round(total_samples / R, 1),
                "step": "press_batch + depress_batch, device resident",
            },
            "ratio": round(ratio, 6),
            "ratio_vs_vbz": round(ratio / VBZ_RATIO, 6),
            "press_MBps": round(raw_bytes / (press_ms * 1e-3) / 1e6, 1),
            "depress_MBps": round(raw_bytes / (depress_ms * 1e-3) / 1e6, 1),
            "roofline": dominant,
            "roofline_other": other,
        }
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(m, sig, starts, n)
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.destroy_process_group()


def measured_traffic(m, reads, seed, fixed_len):
    """HBM bytes per launch of the main kernels from the committed PMC passes
    (profiles/*traffic*.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs,
    gfx950 corrections of MI355X_MICROARCH.md applied) - only for the exact workload they were
    collected on; None otherwise."""
    import glob

    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic*.json")), reverse=True):
        try:
            t = json.load(open(f))
        except (OSError, ValueError):
            continue
        w = t.get("workload", {})
        if (w.get("method"), w.get("reads_per_gpu"), w.get("seed"), w.get("fixed_len")) == (m, reads, seed, fixed_len):
            return t.get("traffic_bytes_per_launch")
    return None


def cpu_baseline(m, sig, starts, n):
    """The reference itself (oracle/_ref, built in the dev container from the reference's
    own sources) or, failing that, the oracle's C restatement, timed on this host with the
    harness's semantics (fresh malloc per read, clock() around X_press / X_depress,
    press/test.c:1756-1815) on a bounded sample of the same reads.  One thread."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _libs

    kind = "reference" if _libs.have_reference() else "port"
    codec = _libs.reference() if kind == "reference" else _libs.oracle()
    k = min(256, len(n))
    host = sig[: int(starts[k])].cpu().numpy()
    parts, off = [], [0]
    for r in range(k):
        parts.append(host[int(starts[r]): int(starts[r]) + int(n[r])])
        off.append(off[-1] + int(n[r]))
    flat = np.concatenate(parts)
    off = np.array(off, dtype=np.uint64)
    raw = 2 * int(off[-1])
    # the reference prints diagnostics on the hot path (press.c:3262)
    devnull = os.open(os.devnull, os.O_WRONLY)
    saved = os.dup(2)
    os.dup2(devnull, 2)
    try:
        ps = ds = 0.0
        passes = 0
        t0 = time.perf_counter()
        while passes < 50 and time.perf_counter() - t0 < 10.0:
            p, d, _ = codec.time_batch(m, flat, off, check=(passes == 0))
            ps += p
            ds += d
            passes += 1
    finally:
        os.dup2(saved, 2)
    ncpu = os.cpu_count()
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {
        "value": round(raw * passes / (ps + ds) / 1e6, 1),
        "unit": "MB/s",
        "cores": 1,
        "kind": kind,
        "sample": "%d reads (%d samples) of the same batch x %d passes; value = raw bytes / (press+depress seconds)"
                  % (k, int(off[-1]), passes),
        "press_MBps": round(raw * passes / ps / 1e6, 1),
        "depress_MBps": round(raw * passes / ds / 1e6, 1),
        "host_cpu": model,
        "host_cores_available": ncpu,
    }


if __name__ == "__main__":
    main()

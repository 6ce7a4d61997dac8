#!/usr/bin/env python3
"""bench.py - the reference's headline measurement on MI355X.

Metric (BASELINE.json): raw-signal MB/s for compress + decompress (MB = 1e6 bytes of
int16 signal, press/test.c's timing of X_press + X_depress) and the compression ratio
against VBZ, on NA12878-like reads.  The NA12878 500k-read set is not available offline,
so the workload is its synthetic stand-in (honours_amd/synth.py: lengths, first samples,
zig-zag-delta symbol statistics and exception rate of the published tables).

One "step" = one pass of the hot path over one device-resident batch of reads:
press_batch then depress_batch.  `value` = raw signal bytes of the batch / step time,
whole job over all ranks (each rank owns its own reads: weak scaling, no data-path
collective; one RCCL all-reduce of the {raw, compressed, reads} totals at the end).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--method M] [--reads R] [--fixed-len L]

Defaults:
  N = 1   headline = BASELINE.json config 4: exception split + static NA12878_zd Huffman
          (shuffman_vbe21_zd, the method that meets "ratio >= VBZ"), 8192 NA12878-like reads.
          The same run then measures config 2 (svb12_zd), config 3 (zstd_svb_zd, full VBZ) and
          config 5's shape on one GPU as sub-records under "configs", each with its own roofline
          and cpu_baseline, and the PCIe-inclusive host-buffer rate ("e2e_host").
  N > 1   BASELINE.json config 5: fixed 200 000-sample reads, best method (shuffman_vbe21_zd),
          4096 reads per GPU per step, launched by the driver with torch.distributed.run
          (one rank per GPU).  No sub-records, no CPU leg.
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
# wave64 VALU issue peak: CDNA4's SIMDs are 32 lanes wide - a wave64 instruction takes the pipe for 2 cycles (one wave
# alone issues every 4: MI355X_MICROARCH.md, 'vector-instruction ISSUE cost'): 256 CUs x 4 SIMDs x 2.4 GHz / 2
VALU_PEAK_GINST = 256 * 4 * 2.4 / 2
HEADLINE = "shuffman_vbe21_zd"
CONFIG5_LEN = 200000
CONFIG5_READS = 4096

WORKLOADS = {
    "svb12_zd": "zig-zag-delta + svb16 pack only (svb12_zd, no entropy stage; BASELINE config 2)",
    "svb_zd": "zig-zag-delta + svb32 pack (svb_zd)",
    "vbe21_zd": "exception split (vbe21_zd)",
    "hasgam_vbsse21_zdq": "ex-zd (hasgam_vbsse21_zdq)",
    "shuffman_vbe21_zd": "exception split + static NA12878_zd Huffman entropy stage (shuffman_vbe21_zd; BASELINE config 4)",
    "slow5_svb_zd": "BLOW5's signal codec (slow5lib svb-zd: u32 count + svb32 of zig-zag deltas)",
    "zstd_svb_zd": "full VBZ pipeline zstd(svb-zd), zstd frames made and read on the device (BASELINE config 3)",
    "zstd_svb12_zd": "zstd(svb16-zd), zstd frames made and read on the device",
    "zstd_hasgam_vbsse21_zdq": "zstd(ex-zd), zstd frames made and read on the device",
    "rc_vbe21_zd": "exception split + order-0 adaptive range coder (rc_vbe21_zd; serial per read by format)",
    "rcc_vbe21_zd": "exception split + order-1 adaptive range coder (rcc_vbe21_zd; serial per read by format)",
    "rccm_vbbe21_zd": "exception split + order 1-0 context-mixing range coder (rccm_vbbe21_zd; serial per read by format)",
}

# (press kernel, depress kernel): the kernels press_hip_kernel_timing() brackets with HIP events
KERNELS = {
    "svb12_zd": ("k_svb_encode_chunked<false,true>", "k_svb_decode_chunked<false,true>"),
    "svb_zd": ("k_svb_encode_chunked<true,true>", "k_svb_decode_chunked<true,true>"),
    "slow5_svb_zd": ("k_svb_encode_chunked<true,true,true>", "k_svb_decode_chunked<true,true,true>"),
    "rc_vbe21_zd": ("k_rcs_encode", "k_rcs_decode"),
    "rcc_vbe21_zd": ("k_rcc_encode", "k_rcc_decode"),
    "rccm_vbbe21_zd": ("k_rcm_encode", "k_rcm_decode"),
    "zstd_svb_zd": ("k_zs_encode", "k_zs_hdecode"),
    "zstd_svb12_zd": ("k_zs_encode", "k_zs_hdecode"),
    "zstd_hasgam_vbsse21_zdq": ("k_zs_encode", "k_zs_hdecode"),
    "shuffman_vbe21_zd": ("k_huff_encode_chunked", "k_huf_sync + repair rounds + k_huf_chain + k_huf_emit"),
}
# what bounds the dominant kernel, from the counter passes kept in profiles/ (DESIGN.md section 4):
# "hbm" = streams at the memory system's rate; "valu" = instruction issue (HBM time of its bytes is a fraction)
LIMITER = {"shuffman_vbe21_zd": "valu", "zstd_svb_zd": "valu", "zstd_svb12_zd": "valu",
           "zstd_hasgam_vbsse21_zdq": "valu", "rc_vbe21_zd": "valu (serial per read)",
           "rcc_vbe21_zd": "valu (serial per read)", "rccm_vbbe21_zd": "valu (serial per read)"}


def kernel_own_bytes(m, which, raw, comp, nsamp):
    """Bytes the timed kernel itself has to move once (its share of the call's 2n + c):
    the kernels of a multi-kernel pipeline hand intermediates to each other through HBM."""
    if m.startswith("shuffman"):
        # encode: samples in, payload out; decode (k_huf_sync .. k_huf_emit, timed together): payload in,
        # samples out (k_huf_emit writes them itself: the one-byte values never reach HBM)
        return raw + comp
    if m.startswith("zstd"):
        # k_zs_encode: inner stream (1.25 B/sample) in, frame out; k_zs_hdecode: frame in, literals out
        inner = nsamp * 5 // 4
        return inner + comp if which == 0 else comp + nsamp
    if m.startswith("rc"):
        return nsamp + comp
    return raw + comp


def measured_traffic(m, reads, seed, fixed_len):
    """HBM bytes per launch from the committed PMC passes (profiles/*traffic*.json: rocprofv3
    --pmc FETCH_SIZE / WRITE_SIZE in separate runs, gfx950 corrections of MI355X_MICROARCH.md
    applied; tools/traffic.sh) - only for the exact workload they were collected on.
    -> (dict or None, file name or None).  Counters cannot be read inside a timed run."""
    import glob

    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic*.json")), reverse=True):
        try:
            t = json.load(open(f))
        except (OSError, ValueError):
            continue
        w = t.get("workload", {})
        if (w.get("method"), w.get("reads_per_gpu"), w.get("seed"), w.get("fixed_len")) == (m, reads, seed, fixed_len):
            return t.get("traffic_bytes_per_launch"), os.path.relpath(f, ROOT)
    return None, None


def measured_valu(m, reads, seed, fixed_len):
    """Wave-level VALU instructions per launch of the method's kernels from the committed counter pass
    (profiles/*valu*.json: rocprofv3 --pmc SQ_INSTS_VALU ..., tools/valu.py) for this exact workload, or None."""
    import glob

    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*valu*.json")), reverse=True):
        try:
            t = json.load(open(f))
        except (OSError, ValueError):
            continue
        w = t.get("workload", {})
        if (w.get("method"), w.get("reads_per_gpu"), w.get("seed"), w.get("fixed_len")) == (m, reads, seed, fixed_len):
            return t.get("valu_insts_per_call"), os.path.relpath(f, ROOT)
    return None, None


class Batch:
    """A device-resident synthetic batch and the arena / index buffers of one method."""

    def __init__(self, torch, press, synth, seed, first_read, R, dev, fixed_len):
        self.R = R
        self.fixed_len = fixed_len
        sig, starts, n = synth.synth_batch_torch(seed, first_read, R, dev, fixed_len=fixed_len, align=64)
        self.sig = torch.cat([sig, torch.zeros(64, dtype=torch.int16, device=dev)])
        self.starts, self.n = starts, n
        self.total_samples = int(n.sum())
        self.raw_bytes = 2 * self.total_samples
        self.d_off = torch.from_numpy(starts[:-1].astype(np.int64)).to(dev)
        self.d_n = torch.from_numpy(n.astype(np.int32)).to(dev)
        self.d_back = torch.zeros_like(self.sig)
        self.d_outn = torch.zeros(R, dtype=torch.int32, device=dev)
        self.d_len = torch.zeros(R, dtype=torch.int64, device=dev)
        self.dev = dev

    def arena(self, torch, press, m):
        caps = np.array([press.bound(m, int(x)) for x in self.n], dtype=np.int64)
        caps = (caps + 64 + 127) // 128 * 128
        out_off = np.concatenate([[0], np.cumsum(caps)])
        d_out = torch.empty(int(out_off[-1]) + 64, dtype=torch.uint8, device=self.dev)
        d_out_off = torch.from_numpy(out_off).to(self.dev)
        return caps, d_out, d_out_off, d_out_off[:-1].contiguous()


def run_method(torch, press, shard, b, m, steps, warmup, world, dist, seed, check=True):
    """warmup, verify, then time EXACTLY `steps` steps between barrier + synchronize pairs.
    -> the record of this method (value over all ranks)."""
    if m.startswith("shuffman"):
        press.load_table()
    caps, d_out, d_out_off, d_in_off = b.arena(torch, press, m)

    def step(ev=None):
        if ev is not None:
            ev[0].record()
        press.press_batch(m, b.sig, b.d_off, b.d_n, d_out, d_out_off, b.d_len)
        if ev is not None:
            ev[1].record()
        press.depress_batch(m, d_out, d_in_off, b.d_len, b.d_back, b.d_off, b.d_n, b.d_outn)
        if ev is not None:
            ev[2].record()

    def barrier():
        if world > 1:
            dist.barrier()

    b.d_back.zero_()
    for _ in range(max(warmup, 1) if check else warmup):
        step()
    torch.cuda.synchronize()
    # correctness of what is being timed: lossless on the device, sizes sane
    lens = b.d_len.cpu().numpy()
    if check:
        assert (lens > 0).all() and (lens < caps).all(), "%s: a read failed to compress" % m
        assert bool((b.d_outn.cpu() == torch.from_numpy(b.n.astype(np.int32))).all()), "%s: sample counts" % m
        assert torch.equal(b.d_back, b.sig), "%s: round trip is not lossless" % m
    comp_bytes = int(lens.sum())

    events = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(steps)]
    press.kernel_timing(True)  # HIP events around the dominant kernel of each call, on its launch stream
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
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
    raw_all, comp_all, reads_all, elapsed, per_rank = shard.gather_totals(b.raw_bytes, comp_bytes, b.R, elapsed, b.dev)

    # Roofline.  Algorithmic bytes of a call = sum over the batch of 2n (int16
    # samples) + c (compressed stream), each crossing HBM once.
    #   call_*   the whole press_batch / depress_batch call (all its kernels): alg / whole_call_ms
    #   achieved/frac  the dominant kernel alone, timed with HIP events on its launch stream, against the
    #            bytes THAT kernel has to move (kernel_bytes) - for a single-kernel method these coincide
    alg = b.raw_bytes + comp_bytes
    kern = KERNELS.get(m, ("k_low_encode_chunked", "k_low_decode_chunked<false>"))
    traffic, tsrc = measured_traffic(m, b.R, seed, b.fixed_len)
    valu, vsrc = measured_valu(m, b.R, seed, b.fixed_len)

    def roof(which, name, ms, call_ms, key):
        own = kernel_own_bytes(m, which, b.raw_bytes, comp_bytes, b.total_samples)
        gbps = own / (ms * 1e-3) / 1e9
        call_gbps = alg / (call_ms * 1e-3) / 1e9
        lim = LIMITER.get(m, "hbm")
        # achieved / peak / frac: always the HBM roofline of the timed kernel (the contract's figure); `bound`
        # names what the counter passes say limits it, and for VALU-bound kernels `valu` is that roofline:
        # wave instructions per launch (counter pass in profiles/) / live kernel time against the issue peak
        r = {"kernel": name, "bound": "valu" if lim.startswith("valu") else "hbm", "achieved": round(gbps, 1),
             "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(gbps / HBM_PEAK_GBPS, 4),
             "traffic": traffic.get(key) if traffic else None,
             "traffic_source": ("from profile %s (counter pass of the same workload, not measured in this run)" % tsrc)
             if tsrc else None,
             "kernel_bytes_per_launch": own, "avg_launch_ms": round(ms, 4),
             "algorithmic_bytes_per_call": alg, "whole_call_ms": round(call_ms, 4),
             "call_achieved": round(call_gbps, 1), "call_frac": round(call_gbps / HBM_PEAK_GBPS, 4),
             "limiter": lim}
        if valu and valu.get(key + "_timed"):
            gi = valu[key + "_timed"] / (ms * 1e-3) / 1e9
            r["valu"] = {"wave_insts_per_launch": valu[key + "_timed"], "achieved": round(gi, 1),
                         "peak": round(VALU_PEAK_GINST, 1), "unit": "G wave-inst/s",
                         "frac": round(gi / VALU_PEAK_GINST, 4),
                         "source": "from profile %s (SQ_INSTS_VALU of the timed kernels) / this run's kernel time" % vsrc}
        return r

    r_press = roof(0, kern[0], kpress_ms, press_ms, "press")
    r_depress = roof(1, kern[1], kdepress_ms, depress_ms, "depress")
    dominant, other = (r_depress, r_press) if kdepress_ms >= kpress_ms else (r_press, r_depress)
    ratio = raw_all / comp_all
    return {
        "value": round(raw_all / (elapsed / steps) / 1e6, 1),
        "unit": "MB/s",
        "ms_per_step": round(elapsed / steps * 1e3, 4),
        "config": {
            "workload": ("synthetic %d-sample reads (BASELINE config 5 shape), " % b.fixed_len if b.fixed_len
                         else "NA12878-like synthetic reads, ") + WORKLOADS[m],
            "method": m,
            "reads_per_gpu": b.R,
            "samples_per_gpu": b.total_samples,
            "mean_read_len": round(b.total_samples / b.R, 1),
            "step": "press_batch + depress_batch, device resident",
        },
        "ratio": round(ratio, 6),
        "ratio_vs_vbz": round(ratio / VBZ_RATIO, 6),
        "press_MBps": round(b.raw_bytes / (press_ms * 1e-3) / 1e6, 1),
        "depress_MBps": round(b.raw_bytes / (depress_ms * 1e-3) / 1e6, 1),
        "roofline": dominant,
        "roofline_other": other,
        # what the one collective brought back: the world RCCL saw and the spread of the ranks' step times
        "ranks": {"world_size": len(per_rank), "backend": "nccl (RCCL)" if world > 1 else "none (one process)",
                  "ms_per_step_min": round(min(r[3] for r in per_rank) / steps * 1e3, 4),
                  "ms_per_step_max": round(max(r[3] for r in per_rank) / steps * 1e3, 4),
                  "reads": reads_all},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--method", default=None, choices=sorted(WORKLOADS),
                    help="headline method (default %s)" % HEADLINE)
    ap.add_argument("--reads", type=int, default=None, help="reads per GPU per step")
    ap.add_argument("--fixed-len", type=int, default=None,
                    help="fixed read length (config 5: %d, the default for --gpus > 1)" % CONFIG5_LEN)
    ap.add_argument("--seed", type=int, default=20261004)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline legs")
    ap.add_argument("--no-sub", action="store_true", help="headline only: no sub-records, no e2e leg")
    ap.add_argument("--no-check", action="store_true", help=argparse.SUPPRESS)  # diagnostic kernel builds only
    ap.add_argument("--full-batches", type=int, default=61,
                    help="config4_500k_equiv: distinct 8192-read batches (61 = 499 712 reads; 0 = skip)")
    ap.add_argument("--full-reps", type=int, default=6, help="config4_500k_equiv: timed steps per batch")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from honours_amd import press, shard, synth

    rank, world, local_rank = shard.world_info()
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d under a world of %d ranks: launch N>1 with python -m "
                         "torch.distributed.run --nproc-per-node N bench.py --gpus N" % (args.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    press.load_library()
    press.use_torch_stream()
    explicit = args.method is not None or args.fixed_len is not None or args.reads is not None
    m = args.method or HEADLINE
    fixed_len = args.fixed_len
    reads = args.reads
    if world > 1 and fixed_len is None and args.reads is None and args.method is None:
        fixed_len, reads = CONFIG5_LEN, CONFIG5_READS  # BASELINE.json config 5
    if reads is None:
        reads = CONFIG5_READS if fixed_len else 8192

    # ---- synthetic batch, generated on the device (never staged through PCIe)
    first_read, R = shard.weak_shard(reads, rank)
    b = Batch(torch, press, synth, args.seed, first_read, R, dev, fixed_len)
    out = run_method(torch, press, shard, b, m, args.steps, args.warmup, world, dist, args.seed,
                     check=not args.no_check)

    if rank == 0:
        line = {
            "metric": "raw-signal MB/s (compress+decompress)",
            "value": out["value"],
            "unit": "MB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": out["ms_per_step"],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u16",
            "data": "synthetic",
        }
        line.update({k: v for k, v in out.items() if k not in ("value", "unit", "ms_per_step")})
        sub = world == 1 and not args.no_sub and not explicit
        if world == 1 and not args.no_cpu:
            line["cpu_baseline"] = cpu_baseline(m, b)
        if sub:
            # the other single-GPU configurations of BASELINE.json, measured the same way in this run
            line["configs"] = {}
            for key, mm in (("config2_svb12_zd", "svb12_zd"), ("config3_zstd_svb_zd", "zstd_svb_zd")):
                rec = run_method(torch, press, shard, b, mm, args.steps, args.warmup, 1, dist, args.seed)
                rec["steps"], rec["warmup"] = args.steps, args.warmup
                if not args.no_cpu:
                    rec["cpu_baseline"] = cpu_baseline(mm, b, budget_s=6.0)
                if mm == "zstd_svb_zd":
                    rec["libzstd_frames"] = libzstd_frames(torch, press, b, args.steps)
                line["configs"][key] = rec
            line["e2e_host"] = e2e_host(torch, press, b, m)
            line["per_read_api"] = per_read_api(press, b, m)
            line["e2e_blow5"] = e2e_blow5(torch, press, b, m)
            del b
            torch.cuda.empty_cache()
            b5 = Batch(torch, press, synth, args.seed, 0, CONFIG5_READS, dev, CONFIG5_LEN)
            rec = run_method(torch, press, shard, b5, m, args.steps, args.warmup, 1, dist, args.seed)
            rec["steps"], rec["warmup"] = args.steps, args.warmup
            line["configs"]["config5_shape_1gpu"] = rec
            del b5
            torch.cuda.empty_cache()
            if args.full_batches > 0:
                line["configs"]["config4_500k_equiv"] = full_scale(torch, press, synth, dev, m, args.seed, reads,
                                                                   args.full_batches, args.full_reps)
        print(json.dumps(line), flush=True)

    if world > 1:
        dist.destroy_process_group()


def full_scale(torch, press, synth, dev, m, seed, reads, nbatches, reps):
    """The metric's own scale (BASELINE: NA12878, 500 000 reads): `nbatches` DISTINCT batches of `reads`
    NA12878-like reads - batch k holds reads [k * reads, (k + 1) * reads) of the same generator, made on the
    device, untimed - each through press_batch + depress_batch `reps` times behind one warm-up step, timed with
    events on the launch stream; every batch checked lossless.  Aggregate ratio and MB/s over all of them, and the
    spread over the batches (the longest reads, 5.7 M samples, sit in the same launch as 2 k ones)."""
    if m.startswith("shuffman"):
        press.load_table()
    tot_raw = tot_comp = tot_reads = 0
    tot_press = tot_depress = 0.0
    per = []
    longest = 0
    t_wall = time.perf_counter()
    for k in range(nbatches):
        b = Batch(torch, press, synth, seed, k * reads, reads, dev, None)
        caps, d_out, d_out_off, d_in_off = b.arena(torch, press, m)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * reps + 1)]
        for it in range(reps + 1):  # the first pass warms up (and sizes the scratch for this batch)
            if it >= 1:
                ev[2 * (it - 1)].record()
            press.press_batch(m, b.sig, b.d_off, b.d_n, d_out, d_out_off, b.d_len)
            if it >= 1:
                ev[2 * (it - 1) + 1].record()
            press.depress_batch(m, d_out, d_in_off, b.d_len, b.d_back, b.d_off, b.d_n, b.d_outn)
        ev[2 * reps].record()
        torch.cuda.synchronize()
        lens = b.d_len.cpu().numpy()
        assert (lens > 0).all() and (lens < caps).all(), "batch %d: a read failed to compress" % k
        assert torch.equal(b.d_back, b.sig), "batch %d: round trip is not lossless" % k
        p_ms = sum(ev[2 * i].elapsed_time(ev[2 * i + 1]) for i in range(reps)) / reps
        d_ms = sum(ev[2 * i + 1].elapsed_time(ev[2 * i + 2]) for i in range(reps)) / reps
        comp = int(lens.sum())
        tot_raw += b.raw_bytes
        tot_comp += comp
        tot_reads += b.R
        tot_press += p_ms
        tot_depress += d_ms
        longest = max(longest, int(b.n.max()))
        per.append((b.raw_bytes / ((p_ms + d_ms) * 1e-3) / 1e6, b.raw_bytes / comp, p_ms, d_ms))
        del b, d_out, d_out_off, d_in_off
    v = [x[0] for x in per]
    return {
        "what": "%d distinct batches x %d NA12878-like reads (reads 0 .. %d of the generator), press_batch + "
                "depress_batch, device resident, every batch checked lossless; %d timed steps per batch behind "
                "one warm-up" % (nbatches, reads, nbatches * reads - 1, reps),
        "method": m, "reads": tot_reads, "samples": tot_raw // 2, "raw_bytes": tot_raw, "compressed_bytes": tot_comp,
        "longest_read": longest,
        "ratio": round(tot_raw / tot_comp, 6), "ratio_vs_vbz": round(tot_raw / tot_comp / VBZ_RATIO, 6),
        "value": round(tot_raw / ((tot_press + tot_depress) * 1e-3) / 1e6, 1), "unit": "MB/s",
        "press_MBps": round(tot_raw / (tot_press * 1e-3) / 1e6, 1),
        "depress_MBps": round(tot_raw / (tot_depress * 1e-3) / 1e6, 1),
        "gpu_ms_timed": round((tot_press + tot_depress) * reps, 1),
        "per_batch": {"value_min": round(min(v), 1), "value_max": round(max(v), 1),
                      "ratio_min": round(min(x[1] for x in per), 6), "ratio_max": round(max(x[1] for x in per), 6),
                      "press_ms_min": round(min(x[2] for x in per), 4), "press_ms_max": round(max(x[2] for x in per), 4),
                      "depress_ms_min": round(min(x[3] for x in per), 4),
                      "depress_ms_max": round(max(x[3] for x in per), 4)},
        "wall_s_incl_generation": round(time.perf_counter() - t_wall, 1),
    }


def per_read_api(press, b, m, nreads=256):
    """the per-read drop-in symbols as press/test.c:1783-1810 calls them - X_press / X_depress of
    ONE read per call, host pointers in and out, the clock around the call alone (buffers allocated outside, as
    the harness does) - on the first `nreads` reads of the batch.  Launch- and PCIe-latency bound by
    construction: the batch API is the throughput path."""
    import ctypes

    lib = press.load_library()
    k = min(nreads, b.R)
    host = b.sig[: int(b.starts[k])].cpu().numpy()
    reads = [np.ascontiguousarray(host[int(b.starts[r]): int(b.starts[r]) + int(b.n[r])]) for r in range(k)]
    _, pname, dname, kind = press._SYMS[m]
    if kind != "shuff":
        return None
    tab = press.default_table()
    fp, fd = getattr(lib, pname), getattr(lib, dname)
    fp.restype = ctypes.c_int
    fp.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
    fd.restype = ctypes.c_int
    fd.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32)]
    outs, backs, lens = [], [], []
    for s in reads:
        outs.append(np.zeros(press.bound(m, s.size) + 1024, dtype=np.uint8))
        backs.append(np.zeros(s.size + 64, dtype=np.int16))
    tp = td = 0.0
    for it in range(2):  # the first pass warms up
        tp = td = 0.0
        lens = []
        for s, o in zip(reads, outs):
            nout = ctypes.c_uint64(o.size)
            t0 = time.perf_counter()
            ret = fp(tab.se, s.ctypes.data, s.size, o.ctypes.data, ctypes.byref(nout))
            tp += time.perf_counter() - t0
            assert ret == 0
            lens.append(int(nout.value))
        for s, o, c, bk in zip(reads, outs, lens, backs):
            nn = ctypes.c_uint32(s.size)
            t0 = time.perf_counter()
            ret = fd(tab.root, o.ctypes.data, c, bk.ctypes.data, ctypes.byref(nn))
            td += time.perf_counter() - t0
            assert ret == 0 and nn.value == s.size
    assert all(np.array_equal(bk[: s.size], s) for s, bk in zip(reads, backs)), "per-read round trip"
    raw = 2 * sum(s.size for s in reads)
    return {"what": "%s_press_16 / _depress_16, one read per call with host pointers (press/test.c:1783-1810), "
                    "the clock around each call" % m,
            "symbols": [pname, dname], "reads": k, "raw_bytes": raw, "unit": "MB/s",
            "press_MBps": round(raw / tp / 1e6, 1), "depress_MBps": round(raw / td / 1e6, 1),
            "value": round(raw / (tp + td) / 1e6, 1),
            "press_us_per_call": round(tp / k * 1e6, 1), "depress_us_per_call": round(td / k * 1e6, 1)}


def e2e_blow5(torch, press, b, m, nreads=2048, batch_reads=512):
    """The pipeline the reference's SLOW5/BLOW5 read iterator feeds, file to compressed streams on the
    host: a BLOW5 file (zlib records, svb-zd signals: what slow5tools writes) -> press_hip_blow5_next (records
    inflated by a pool of host threads, signal fields AS STORED into page-locked memory) -> device -> svb-zd decode
    -> `m` -> streams back on the host.  The file holds the first `nreads` reads of the batch (made here, untimed,
    with the library's own writer).  PCIe and file reading included: never `value`."""
    import queue
    import threading

    like = os.path.join(ROOT, "tests", "golden", "three-reads.blow5")
    if not os.path.exists(like):
        return None
    k = min(nreads, b.R)
    n = b.n[:k]
    # ---- the file (untimed): the device's svb-zd streams of the reads, framed and deflated by the writer
    d_off, d_n = b.d_off[:k].contiguous(), b.d_n[:k].contiguous()
    caps = (np.array([press.bound("slow5_svb_zd", int(x)) for x in n], dtype=np.int64) + 64 + 127) // 128 * 128
    out_off = np.concatenate([[0], np.cumsum(caps)])
    d_tmp = torch.empty(int(out_off[-1]) + 64, dtype=torch.uint8, device=b.dev)
    d_len = torch.zeros(k, dtype=torch.int64, device=b.dev)
    press.press_batch("slow5_svb_zd", b.sig, d_off, d_n, d_tmp, torch.from_numpy(out_off).to(b.dev), d_len)
    torch.cuda.synchronize()
    host, lens = d_tmp.cpu().numpy(), d_len.cpu().numpy()
    fields = [host[int(out_off[r]): int(out_off[r]) + int(lens[r])] for r in range(k)]
    path = "/tmp/press_hip_bench_%d.blow5" % os.getpid()
    press.blow5_write_like(path, like, fields, record_method=1, signal_method=1)
    del d_tmp, host, fields
    file_bytes = os.path.getsize(path)
    raw = 2 * int(n.sum())
    if m.startswith("shuffman"):
        press.load_table()
    arena_cap = 1 << 30
    try:
        # ---- the reader alone (file -> inflated signal fields in page-locked memory)
        pin = [torch.empty(arena_cap, dtype=torch.uint8).pin_memory() for _ in range(2)]
        meta = [(np.zeros(batch_reads, np.uint64), np.zeros(batch_reads, np.uint64), np.zeros(batch_reads, np.uint32))
                for _ in range(2)]

        def reader_pass(consume):
            rd = press.Blow5Reader(path)
            q = queue.Queue(maxsize=1)

            def produce():
                slot = 0
                while True:
                    off, ln, ns = meta[slot]
                    got = rd.next_arena(pin[slot].numpy(), off, ln, ns, batch_reads)
                    q.put((slot, got))
                    if got == 0:
                        return
                    slot ^= 1

            th = threading.Thread(target=produce)
            th.start()
            reads = 0
            while True:
                slot, got = q.get()
                if got == 0:
                    break
                consume(slot, got)
                reads += got
            th.join()
            rd.close()
            return reads

        t0 = time.perf_counter()
        assert reader_pass(lambda slot, got: None) == k
        t_reader = time.perf_counter() - t0

        # ---- the whole way
        total_comp = [0]
        check = []

        def gpu_batch(slot, got):
            off, ln, ns = meta[slot]
            used = int(off[got - 1] + ln[got - 1])
            d_in = pin[slot][:used + 64].to(b.dev, non_blocking=True)
            nsv = ns[:got].astype(np.int64)
            starts = np.concatenate([[0], np.cumsum((nsv + 63) // 64 * 64)])
            d_in_off = torch.from_numpy(off[:got].astype(np.int64)).to(b.dev)
            d_in_len = torch.from_numpy(ln[:got].astype(np.int64)).to(b.dev)
            d_soff = torch.from_numpy(starts[:-1].copy()).to(b.dev)
            d_ns = torch.from_numpy(nsv.astype(np.int32)).to(b.dev)
            d_sig = torch.empty(int(starts[-1]) + 64, dtype=torch.int16, device=b.dev)
            d_outn = torch.zeros(got, dtype=torch.int32, device=b.dev)
            press.depress_batch("slow5_svb_zd", d_in, d_in_off, d_in_len, d_sig, d_soff, d_ns, d_outn)
            cp = (np.array([press.bound(m, int(x)) for x in nsv], dtype=np.int64) + 64 + 127) // 128 * 128
            oo = np.concatenate([[0], np.cumsum(cp)])
            d_out = torch.empty(int(oo[-1]) + 64, dtype=torch.uint8, device=b.dev)
            d_l = torch.zeros(got, dtype=torch.int64, device=b.dev)
            press.press_batch(m, d_sig, d_soff, d_ns, d_out, torch.from_numpy(oo).to(b.dev), d_l)
            h_out = d_out.cpu()  # (the streams cross the link in their slots; lengths beside them)
            h_len = d_l.cpu().numpy()
            assert (h_len > 0).all() and bool((d_outn.cpu().numpy() == nsv).all())
            total_comp[0] += int(h_len.sum())
            if not check:  # the first batch against the batch's own samples
                first = d_sig[: int(nsv[0])].cpu()
                check.append(bool(torch.equal(first, b.sig[int(b.starts[0]): int(b.starts[0]) + int(nsv[0])].cpu())))
            del h_out

        reader_pass(gpu_batch)  # warm-up (scratch, allocator)
        total_comp[0] = 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        assert reader_pass(gpu_batch) == k
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        assert check and check[0], "BLOW5 e2e: the decoded samples differ"
    finally:
        os.remove(path)
    return {"what": "BLOW5 file (zlib records, svb-zd signals) -> press_hip_blow5_next on a pool of host threads -> "
                    "page-locked arena -> device: slow5_svb_zd decode -> %s -> streams on the host; %d reads per batch, "
                    "the reader one batch ahead of the device" % (m, batch_reads),
            "reads": k, "raw_bytes": raw, "file_bytes": file_bytes, "compressed_bytes": total_comp[0],
            "ratio_vs_file": round(file_bytes / total_comp[0], 4), "unit": "MB/s",
            "value": round(raw / t_all / 1e6, 1), "reader_alone_MBps": round(raw / t_reader / 1e6, 1),
            "file_MBps": round(file_bytes / t_all / 1e6, 1), "host_threads": min(os.cpu_count() or 1, 32)}


def libzstd_frames(torch, press, b, steps, nreads=1024):
    """Config 3 the other way round: the REFERENCE's VBZ streams - ZSTD_compress level 1 of [u32 n][svb-zd]
    (press.c:1860, press.h:275), made here by the host's libzstd from the device's own svb-zd streams - read
    by the device (sequences, FSE tables, match copies: k_zs_exec).  `zstd_host_frames` = frames the device
    handed to libzstd on the host (0 = none).  None if this host has no libzstd."""
    import ctypes
    import struct

    z = press.open_libzstd()  # (the copy already in the process first: two versions side by side abort)
    if z is None:
        return None
    k = min(nreads, b.R)
    n = b.n[:k]
    d_off, d_n = b.d_off[:k].contiguous(), b.d_n[:k].contiguous()
    caps = (np.array([press.bound("svb_zd", int(x)) for x in n], dtype=np.int64) + 64 + 127) // 128 * 128
    out_off = np.concatenate([[0], np.cumsum(caps)])
    d_out = torch.empty(int(out_off[-1]) + 64, dtype=torch.uint8, device=b.dev)
    d_out_off = torch.from_numpy(out_off).to(b.dev)
    d_len = torch.zeros(k, dtype=torch.int64, device=b.dev)
    press.press_batch("svb_zd", b.sig, d_off, d_n, d_out, d_out_off, d_len)
    torch.cuda.synchronize()
    host, lens = d_out.cpu().numpy(), d_len.cpu().numpy()
    frames = []
    for r in range(k):
        inner = struct.pack("<I", int(n[r])) + host[int(out_off[r]): int(out_off[r]) + int(lens[r])].tobytes()
        src = np.frombuffer(inner, dtype=np.uint8)
        dst = np.empty(len(inner) + len(inner) // 100 + 1024, dtype=np.uint8)
        c = z.ZSTD_compress(dst.ctypes.data, dst.size, src.ctypes.data, len(inner), 1)
        assert not z.ZSTD_isError(c)
        frames.append(dst[:c].tobytes())
    flen = np.array([len(f) for f in frames], dtype=np.int64)
    foff = np.concatenate([[0], np.cumsum((flen + 63) // 64 * 64)])
    arena = np.zeros(int(foff[-1]) + 64, dtype=np.uint8)
    for r, f in enumerate(frames):
        arena[int(foff[r]): int(foff[r]) + len(f)] = np.frombuffer(f, dtype=np.uint8)
    d_in = torch.from_numpy(arena).to(b.dev)
    d_in_off = torch.from_numpy(foff[:-1].copy()).to(b.dev)
    d_in_len = torch.from_numpy(flen).to(b.dev)
    d_outn = torch.zeros(k, dtype=torch.int32, device=b.dev)
    b.d_back.zero_()
    nsamp = int(b.starts[k])
    back = b.d_back[:nsamp]  # (the call sizes its scratch and its grids by the samples it is given: these k reads', not the batch's)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for it in range(1 + steps):  # the first pass warms up
        if it == 1:
            ev[0].record()
        press.depress_batch("zstd_svb_zd", d_in, d_in_off, d_in_len, back, d_off, d_n, d_outn)
    ev[1].record()
    torch.cuda.synchronize()
    assert torch.equal(b.d_back[:nsamp], b.sig[:nsamp]), "libzstd frames: not lossless"
    ms = ev[0].elapsed_time(ev[1]) / steps
    raw = 2 * int(n.sum())
    return {"what": "depress_batch(zstd_svb_zd) of ZSTD_compress level-1 frames (the reference's own VBZ streams), "
                    "device resident", "reads": k, "raw_bytes": raw, "ratio": round(raw / float(flen.sum()), 6),
            "depress_ms": round(ms, 4), "depress_MBps": round(raw / (ms * 1e-3) / 1e6, 1),
            "zstd_host_frames": int(press.load_library().press_hip_zstd_host_frames())}


def e2e_host(torch, press, b, m, nreads=2048):
    """PCIe-inclusive rates - never `value`: the batch API with HOST buffers (device_resident = 0:
    H2D of the samples, kernels, D2H of the streams; and back) on the first `nreads` reads of the
    batch, from pageable memory and from page-locked memory (press_hip_host_alloc)."""
    k = min(nreads, b.R)
    host = b.sig[: int(b.starts[k])].cpu().numpy()
    reads = [host[int(b.starts[r]): int(b.starts[r]) + int(b.n[r])] for r in range(k)]
    raw = 2 * int(sum(len(r) for r in reads))
    out = {"method": m, "reads": k, "raw_bytes": raw, "unit": "MB/s",
           "what": "press_hip_press_batch / _depress_batch with host pointers (device_resident = 0), best of 3; "
                   "includes H2D of the samples, D2H of the streams and back"}
    for key, pinned in (("pageable", False), ("pinned", True)):
        hb = press.HostBatch(m, reads, pinned=pinned)
        best_p = best_d = 1e30
        for _ in range(4):  # the first pass allocates scratch and staging
            t0 = time.perf_counter()
            hb.press()
            best_p = min(best_p, time.perf_counter() - t0)
        for _ in range(4):
            t0 = time.perf_counter()
            hb.depress()
            best_d = min(best_d, time.perf_counter() - t0)
        assert hb.lossless(), "e2e round trip"
        hb.close()
        out[key] = {"press_MBps": round(raw / best_p / 1e6, 1), "depress_MBps": round(raw / best_d / 1e6, 1),
                    "value": round(raw / (best_p + best_d) / 1e6, 1)}
    return out


def cpu_baseline(m, b, budget_s=10.0):
    """The reference itself (oracle/_ref, built in the dev container from the reference's
    own sources) or, failing that, the oracle's C restatement, timed on this host with the
    harness's semantics (fresh malloc per read, clock() around X_press / X_depress,
    press/test.c:1756-1815) on a bounded sample of the same reads.  One thread (`value`), and
    the same reads dealt out over the host threads this process may use (`all_cores`)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _libs

    kind = "reference" if _libs.have_reference() else "port"
    codec = _libs.reference() if kind == "reference" else _libs.oracle()
    k = min(256, b.R)
    host = b.sig[: int(b.starts[k])].cpu().numpy()
    parts, off = [], [0]
    for r in range(k):
        parts.append(host[int(b.starts[r]): int(b.starts[r]) + int(b.n[r])])
        off.append(off[-1] + int(b.n[r]))
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
        while passes < 50 and time.perf_counter() - t0 < budget_s:
            p, d, _ = codec.time_batch(m, flat, off, check=(passes == 0))
            ps += p
            ds += d
            passes += 1
        all_cores = cpu_all_cores(codec, m, flat, off, budget_s / 2)
    finally:
        os.dup2(saved, 2)
        os.close(devnull)
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
        "all_cores": all_cores,
        "host_cpu": model,
        "host_cores_available": os.cpu_count(),
    }


def cpu_all_cores(codec, m, flat, off, budget_s):
    """The same reads, one slice per host thread (reads are independent: BASELINE.md 3b), wall
    clock around the pool: raw bytes / seconds of press + depress together.  Threads = the CPUs
    this process may run on, capped at 64."""
    from concurrent.futures import ThreadPoolExecutor

    try:
        ncpu = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        ncpu = os.cpu_count() or 1
    nthr = max(1, min(ncpu, 64, len(off) - 1))
    k = len(off) - 1
    # deal the reads out so that every thread gets about the same number of samples
    order = np.argsort(-(off[1:] - off[:-1]).astype(np.int64))
    bins = [[] for _ in range(nthr)]
    load = [0] * nthr
    for r in order:
        i = int(np.argmin(load))
        bins[i].append(int(r))
        load[i] += int(off[r + 1] - off[r])
    jobs = []
    for rs in bins:
        parts = [flat[int(off[r]): int(off[r + 1])] for r in rs]
        o = np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.uint64)
        jobs.append((np.concatenate(parts) if parts else np.zeros(0, np.int16), o))

    def work(j):
        if len(j[1]) > 1:
            codec.time_batch(m, j[0], j[1], check=False)

    passes, t0 = 0, time.perf_counter()
    with ThreadPoolExecutor(nthr) as ex:
        while passes < 20 and time.perf_counter() - t0 < budget_s:
            list(ex.map(work, jobs))
            passes += 1
    dt = time.perf_counter() - t0
    raw = 2 * int(off[-1])
    return {"value": round(raw * passes / dt / 1e6, 1), "unit": "MB/s", "cores": nthr,
            "sample": "the same %d reads dealt out over %d threads x %d passes, wall clock" % (k, nthr, passes)}


if __name__ == "__main__":
    main()

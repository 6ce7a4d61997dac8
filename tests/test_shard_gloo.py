"""CPU, world_size 2 over gloo: the N > 1 path of bench.py - disjoint read shards, the
totals all-reduce and the max-over-ranks time - with the oracle standing in for the
kernels (the data path itself has no collective)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker_by_samples(rank, world, port, total_reads, q):
    """world_size 4, heavy-tailed read lengths, shards balanced by SAMPLES (shard_range with lengths)"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _libs
    from honours_amd import shard, synth

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, first_samples = synth.read_lengths(91, 0, total_reads)
    n = np.minimum(n, 40000)  # (the oracle is the stand-in for the kernels: keep it to seconds)
    first, count = shard.shard_range(total_reads, rank, world, lengths=n)
    o = _libs.oracle()
    raw = comp = 0
    for k in range(first, first + count):
        s = synth.synth_read(91, k, int(n[k]), int(first_samples[k]))
        ret, c = o.press("hasgam_vbsse21_zdq", s)
        assert ret == 0
        raw += 2 * s.size
        comp += len(c)
    tot = shard.gather_totals(raw, comp, count, 0.5 + 0.25 * rank)
    q.put((rank, first, count, raw, tot))
    dist.barrier()
    dist.destroy_process_group()


def _worker(rank, world, port, total_reads, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _libs
    from honours_amd import shard, synth

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, count = shard.shard_range(total_reads, rank, world)
    sig, off = synth.synth_batch(77, first, count, fixed_len=3000)
    o = _libs.oracle()
    comp = 0
    for k in range(count):
        ret, c = o.press("svb12_zd", sig[int(off[k]):int(off[k + 1])])
        assert ret == 0
        comp += len(c)
    raw, comp_all, nreads, t = shard.reduce_totals(2 * int(off[-1]), comp, count, 1.0 + rank)
    q.put((rank, first, count, raw, comp_all, nreads, t))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_reduce():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _libs
    from honours_amd import shard, synth

    total_reads, world = 7, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total_reads, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # shards are disjoint and cover the read index
    assert [(r[1], r[2]) for r in res] == [(0, 4), (4, 3)]
    # every rank holds the same global totals, equal to a single-process run
    sig, off = synth.synth_batch(77, 0, total_reads, fixed_len=3000)
    o = _libs.oracle()
    comp = sum(len(o.press("svb12_zd", sig[int(off[k]):int(off[k + 1])])[1]) for k in range(total_reads))
    for r in res:
        assert r[3:6] == (2 * int(off[-1]), comp, total_reads)
        assert r[6] == 2.0  # MAX over ranks of (1.0 + rank)


def test_four_rank_shard_by_samples_and_gather():
    """world_size 4 on heavy-tailed lengths: shards by sample totals, ONE all-gather carries sums, the slowest
    rank's time and every rank's own record"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from honours_amd import shard, synth

    total_reads, world = 48, 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_by_samples, args=(r, world, port, total_reads, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n, _ = synth.read_lengths(91, 0, total_reads)
    n = np.minimum(n, 40000)
    # contiguous, disjoint, complete; about equal samples
    assert res[0][1] == 0 and sum(r[2] for r in res) == total_reads
    for a, b in zip(res, res[1:]):
        assert a[1] + a[2] == b[1]
    per = [r[3] for r in res]
    assert max(abs(x - 2 * int(n.sum()) / world) for x in per) <= 2 * int(n.max())
    for r in res:
        raw, comp, reads, tmax, rows = r[4]
        assert raw == 2 * int(n.sum()) and reads == total_reads and comp == sum(x[1] for x in rows)
        assert tmax == 1.25 and [x[3] for x in rows] == [0.5, 0.75, 1.0, 1.25]
        assert [x[0] for x in rows] == per and [x[2] for x in rows] == [q[2] for q in res]


def test_shard_range_balanced():
    from honours_amd import shard

    for total in (0, 1, 7, 8, 500000):
        for world in (1, 2, 3, 8):
            parts = [shard.shard_range(total, r, world) for r in range(world)]
            assert parts[0][0] == 0 and sum(c for _, c in parts) == total
            for a, b in zip(parts, parts[1:]):
                assert a[0] + a[1] == b[0]
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1


def test_shard_range_by_samples():
    """contiguous ranges of the read index with about equal sample totals - read
    lengths are heavy tailed, so the cut is placed on the running sum of the lengths."""
    from honours_amd import shard, synth

    n, _ = synth.read_lengths(5, 0, 4000)
    for world in (1, 2, 3, 4, 8):
        parts = [shard.shard_range(n.size, r, world, lengths=n) for r in range(world)]
        assert parts[0][0] == 0 and sum(c for _, c in parts) == n.size
        for a, b in zip(parts, parts[1:]):
            assert a[0] + a[1] == b[0]
        tot = [int(n[f:f + c].sum()) for f, c in parts]
        # no rank is further from its fair share than one (longest) read
        assert max(abs(t - int(n.sum()) / world) for t in tot) <= int(n.max())
        # and by read COUNT the same split would be visibly unequal in samples
    # degenerate inputs
    assert shard.shard_range(0, 0, 2, lengths=[]) == (0, 0)
    assert shard.shard_range(1, 0, 2, lengths=[10]) in ((0, 1), (0, 0))
    assert sum(shard.shard_range(1, r, 2, lengths=[10])[1] for r in range(2)) == 1
    # one giant read next to small ones: it gets a rank of its own
    lens = [5_000_000] + [10_000] * 500
    p = [shard.shard_range(len(lens), r, 2, lengths=lens) for r in range(2)]
    assert p[0] == (0, 1) and p[1] == (1, 500)

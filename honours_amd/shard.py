"""Multi-GPU sharding of the per-read hot path.

Reads are independent (thesis/probspace/focus.tex:177-183), so the path shards with no
data exchange: rank k of W owns a contiguous range of the read index; every rank runs
the same kernels on its own reads, one process per GPU.  The only collective is ONE
all-gather of {raw bytes, compressed bytes, reads, elapsed microseconds} per rank (32 bytes
each; RCCL over xGMI on GPUs, gloo in the CPU tests): sums, the slowest rank's time and the
spread over the ranks all come out of that one exchange.
"""
import os


def world_info():
    """(rank, world_size, local_rank) from the torch.distributed.run environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def shard_range(total_reads, rank, world, lengths=None):
    """Contiguous split of [0, total_reads) - rank's [first, first+count).

    lengths (samples per read): ranks get about equal SAMPLE totals - read lengths are heavy
    tailed (2 k ... 5.7 M samples, thesis/plots/n-tab.tex), so equal read counts would leave
    the ranks with unequal work.  Cut k sits where the running sum of the lengths first
    reaches k/world of the total; every rank computes the same cuts from the same lengths.
    Without lengths (equal-length reads, BASELINE.json config 5): equal read counts."""
    if lengths is None:
        base, rem = divmod(total_reads, world)
        first = rank * base + min(rank, rem)
        return first, base + (1 if rank < rem else 0)
    import numpy as np

    lengths = np.asarray(lengths, dtype=np.int64)
    assert lengths.size == total_reads
    cum = np.cumsum(lengths)
    total = int(cum[-1]) if total_reads else 0
    # reads [cut[k], cut[k+1]) go to rank k: a read belongs to the rank its MIDPOINT falls into
    mid = cum - lengths / 2.0
    cuts = [int(np.searchsorted(mid, total * k / world, side="left")) for k in range(world)] + [total_reads]
    return cuts[rank], cuts[rank + 1] - cuts[rank]


def weak_shard(reads_per_rank, rank):
    """Weak scaling (bench.py): every rank owns `reads_per_rank` reads of its own."""
    return rank * reads_per_rank, reads_per_rank


def gather_totals(raw_bytes, comp_bytes, nreads, elapsed_s, device=None):
    """THE collective of the path: one all-gather of {raw bytes, compressed bytes, reads, elapsed
    microseconds} per rank - 32 bytes each, RCCL over xGMI on GPUs, gloo in the CPU tests.  Every rank gets
    every rank's record, so sums, the slowest rank's time (what bench.py divides by) and the spread over the
    ranks all come from this one exchange.
    -> (raw, comp, reads, max elapsed seconds, per-rank list of (raw, comp, reads, seconds))."""
    import torch
    import torch.distributed as dist

    mine = (int(raw_bytes), int(comp_bytes), int(nreads), int(round(float(elapsed_s) * 1e6)))
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return mine[0], mine[1], mine[2], mine[3] / 1e6, [(mine[0], mine[1], mine[2], mine[3] / 1e6)]
    world = dist.get_world_size()
    send = torch.tensor(mine, dtype=torch.int64, device=device)
    recv = torch.empty(4 * world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(recv, send)
    rows = recv.view(world, 4).tolist()
    per_rank = [(int(r[0]), int(r[1]), int(r[2]), r[3] / 1e6) for r in rows]
    return (sum(r[0] for r in per_rank), sum(r[1] for r in per_rank), sum(r[2] for r in per_rank),
            max(r[3] for r in per_rank), per_rank)


def reduce_totals(raw_bytes, comp_bytes, nreads, elapsed_s, device=None):
    """Sum of the byte / read totals and the slowest rank's time (gather_totals without the per-rank rows)."""
    return gather_totals(raw_bytes, comp_bytes, nreads, elapsed_s, device)[:4]

"""Synthetic NA12878-like raw nanopore reads.

The NA12878 500k-read set is not available offline, so configs 2-5 of BASELINE.json
run on reads generated from the statistics the reference publishes:

* read lengths: log-normal with median 80 305 and mean 113 471 samples, clamped to
  [2 024, 5 724 000] (thesis/plots/n-tab.tex);
* first sample ~ N(475, 35) clamped to [158, 1748] (thesis/plots/rawsig-tab.tex);
* one-byte zig-zag-delta symbols i.i.d. from the empirical NA12878 table
  ``NA12878_zd_freq`` (press/gen_huffman.c:11; entropy 5.39 bit), exceptions
  (zig-zag delta > 255) injected at 4.8546 per 113 471 samples
  (thesis/plots/ex-tab.tex:14);
* the free walk of those deltas is folded (mirror boundaries) into the 11-bit ADC
  range [0, 2047] (digitisation 2048, thesis/plots/data-meta.tex:11).

The generator is counter based - sample i of read r depends only on (seed, r, i) -
so the numpy implementation here (tests, fixtures, CPU baseline sample) and the
torch implementation (bench.py, on device, never staged through PCIe) produce the
same reads bit for bit.  This module is bench/test plumbing, not the hot path.
"""
import json
import math
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_MASK64 = (1 << 64) - 1
_GAMMA = 0x9E3779B97F4A7C15
_M1 = 0xBF58476D1CE4E5B9
_M2 = 0x94D049BB133111EB

N_MIN, N_MAX = 2024, 5_724_000
LN_MU = math.log(80305.0)
LN_SIGMA = math.sqrt(2.0 * math.log(113471.0 / 80305.0))
# P(exception) per sample, as a threshold on 24 random bits
EXC_THRESH = round(4.8546 / 113471.0 * (1 << 24))
FOLD = 4094  # 2 * 2047


def _cdf32():
    """cumulative NA12878_zd_freq scaled to 2**32 (exact integer arithmetic)."""
    with open(os.path.join(_HERE, "data", "NA12878_zd_freq.json")) as fp:
        freq = json.load(fp)["freq"]
    total = sum(freq)
    cum, acc = [], 0
    for f in freq:
        acc += f
        cum.append((acc << 32) // total)
    cum[-1] = 1 << 32
    return np.array(cum, dtype=np.uint64)


_CDF = None


def cdf32():
    global _CDF
    if _CDF is None:
        _CDF = _cdf32()
    return _CDF


def _splitmix(x):
    """splitmix64 finaliser over a uint64 numpy array (wraps mod 2**64)."""
    with np.errstate(over="ignore"):
        z = x + np.uint64(_GAMMA)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(_M1)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(_M2)
        return z ^ (z >> np.uint64(31))


def _splitmix_int(x):
    z = (x + _GAMMA) & _MASK64
    z = ((z ^ (z >> 30)) * _M1) & _MASK64
    z = ((z ^ (z >> 27)) * _M2) & _MASK64
    return z ^ (z >> 31)


def read_lengths(seed, first_read, nreads, fixed_len=None):
    """Per-read (length, first sample) - generated on the host for both back ends."""
    if fixed_len is not None:
        n = np.full(nreads, int(fixed_len), dtype=np.int64)
    else:
        n = np.empty(nreads, dtype=np.int64)
    first = np.empty(nreads, dtype=np.int64)
    for k in range(nreads):
        r = first_read + k
        h1 = _splitmix_int((seed * 0xD1342543DE82EF95 + r * 4 + 1) & _MASK64)
        h2 = _splitmix_int((seed * 0xD1342543DE82EF95 + r * 4 + 2) & _MASK64)
        h3 = _splitmix_int((seed * 0xD1342543DE82EF95 + r * 4 + 3) & _MASK64)
        u1 = ((h1 >> 11) + 1) / float(1 << 53)
        u2 = (h2 >> 11) / float(1 << 53)
        u3 = ((h3 >> 11) + 1) / float(1 << 53)
        g1 = math.sqrt(-2.0 * math.log(u1)) * math.cos(2.0 * math.pi * u2)
        g2 = math.sqrt(-2.0 * math.log(u3)) * math.sin(2.0 * math.pi * u2)
        if fixed_len is None:
            n[k] = min(max(int(math.exp(LN_MU + LN_SIGMA * g1)), N_MIN), N_MAX)
        first[k] = min(max(int(round(475.0 + 35.0 * g2)), 158), 1748)
    return n, first


def _zvalues_np(seed, r, n):
    """zig-zag-delta values (uint16) of samples 1..n-1 of read r, plus a dummy at 0."""
    i = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = (np.uint64((seed * 0x2545F4914F6CDD1D) & _MASK64) ^ (np.uint64(r) << np.uint64(32))) + i * np.uint64(2)
    h1 = _splitmix(x)
    h2 = _splitmix(x + np.uint64(1))
    sym = np.searchsorted(cdf32(), h1 >> np.uint64(32), side="right").astype(np.int64)
    is_ex = (h1 & np.uint64(0xFFFFFF)) < np.uint64(EXC_THRESH)
    big = ((h2 >> np.uint64(60)) == 0)
    small_v = (h2 & np.uint64(0xFFFFFFFF)) % np.uint64(43)
    big_v = (h2 & np.uint64(0xFFFFFFFF)) % np.uint64(2062)
    exv = 256 + np.where(big, big_v, small_v).astype(np.int64)
    return np.where(is_ex, exv, sym)


def synth_read(seed, r, n, first):
    """One read as int16[n] (numpy reference implementation of the generator)."""
    z = _zvalues_np(seed, r, int(n))
    delta = (z >> 1) ^ -(z & 1)
    delta[0] = int(first)
    w = np.cumsum(delta)
    m = np.mod(w, FOLD)
    return np.where(m <= 2047, m, FOLD - m).astype(np.int16)


def synth_batch(seed, first_read, nreads, fixed_len=None):
    """-> (signal int16[total], offsets uint64[nreads+1]) on the host."""
    n, first = read_lengths(seed, first_read, nreads, fixed_len)
    off = np.zeros(nreads + 1, dtype=np.uint64)
    off[1:] = np.cumsum(n)
    sig = np.empty(int(off[-1]), dtype=np.int16)
    for k in range(nreads):
        sig[int(off[k]): int(off[k + 1])] = synth_read(seed, first_read + k, n[k], first[k])
    return sig, off


# ---------------------------------------------------------------------------- torch back end

def _i64(c):
    """python int (mod 2**64) -> signed 64-bit value usable as a torch int64 scalar."""
    c &= _MASK64
    return c - (1 << 64) if c >= (1 << 63) else c


def _lsr(t, s):
    """logical shift right of an int64 tensor."""
    return (t >> s) & ((1 << (64 - s)) - 1)


def _splitmix_t(x):
    z = x + _i64(_GAMMA)
    z = (z ^ _lsr(z, 30)) * _i64(_M1)
    z = (z ^ _lsr(z, 27)) * _i64(_M2)
    return z ^ _lsr(z, 31)


def synth_batch_torch(seed, first_read, nreads, device, fixed_len=None, group_samples=1 << 27, align=1):
    """Same reads as synth_batch, generated on `device`.

    -> (int16 tensor, starts uint64 ndarray [nreads+1], n int64 ndarray [nreads]).
    Read k occupies sig[starts[k] : starts[k] + n[k]]; with align > 1 every start is a
    multiple of `align` samples (the gaps are zero) and starts[nreads] is the padded
    total; with align == 1 the layout equals synth_batch's."""
    import torch

    n, first = read_lengths(seed, first_read, nreads, fixed_len)
    pad = (n + align - 1) // align * align
    starts = np.zeros(nreads + 1, dtype=np.uint64)
    starts[1:] = np.cumsum(pad)
    total = int(starts[-1])
    sig = torch.zeros(total, dtype=torch.int16, device=device)
    cdf = torch.from_numpy(cdf32().astype(np.int64)).to(device)
    seedmul = (seed * 0x2545F4914F6CDD1D) & _MASK64
    k0 = 0
    while k0 < nreads:
        k1 = k0 + 1
        while k1 < nreads and int(n[k0:k1 + 1].sum()) <= group_samples:
            k1 += 1
        g_n = torch.from_numpy(n[k0:k1]).to(device)
        g_tot = int(n[k0:k1].sum())
        g_off_np = np.zeros(k1 - k0, dtype=np.int64)
        g_off_np[1:] = np.cumsum(n[k0:k1])[:-1]
        g_off = torch.from_numpy(g_off_np).to(device)
        rid = torch.repeat_interleave(torch.arange(k0, k1, device=device, dtype=torch.int64), g_n)
        idx = torch.arange(g_tot, device=device, dtype=torch.int64) - g_off[rid - k0]
        x = (_i64(seedmul) ^ ((rid + first_read) << 32)) + idx * 2
        h1 = _splitmix_t(x)
        h2 = _splitmix_t(x + 1)
        sym = torch.searchsorted(cdf, _lsr(h1, 32), right=True)
        is_ex = (h1 & 0xFFFFFF) < EXC_THRESH
        low = h2 & 0xFFFFFFFF
        exv = 256 + torch.where(_lsr(h2, 60) == 0, low % 2062, low % 43)
        z = torch.where(is_ex, exv, sym)
        del h1, h2, sym, is_ex, low, exv, x
        delta = (z >> 1) ^ -(z & 1)
        del z
        delta[g_off] = torch.from_numpy(first[k0:k1]).to(device)
        w = torch.cumsum(delta, 0)
        # subtract the walk accumulated by the previous reads of this group
        base = torch.zeros(k1 - k0, dtype=torch.int64, device=device)
        if k1 - k0 > 1:
            base[1:] = w[g_off[1:] - 1]
        w -= base[rid - k0]
        del delta
        m = torch.remainder(w, FOLD)
        vals = torch.where(m <= 2047, m, FOLD - m).to(torch.int16)
        del w, m
        if align == 1:
            sig[int(starts[k0]): int(starts[k0]) + g_tot] = vals
        else:
            dst = torch.from_numpy(starts[k0:k1].astype(np.int64)).to(device)[rid - k0] + idx
            sig[dst] = vals
            del dst
        del vals, rid, idx
        k0 = k1
    return sig, starts, n

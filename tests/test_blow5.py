"""BLOW5 input row: the product's BLOW5 reader and the slow5lib "svb-zd" signal codec
against the reference's own fixture data/three-reads.blow5 (committed as tests/golden/three-reads.blow5,
written by slow5tools with record = zlib, signal = svb-zd) and the samples the reference's loader
dumps from it (tests/golden/three_reads.i16.bin, made by oracle/ref_dump_blow5 -> make_golden.py).

The signal bytes in the file are real known answers for the codec: nobody in this repository
produced them."""
import json
import os
import struct
import zlib

import numpy as np
import pytest

import _libs
from honours_amd import press

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BLOW5 = os.path.join(GOLD, "three-reads.blow5")


def parse_blow5_py(path):
    """independent parse of the file (the layout of slow5.c:787-870, 3903-3965), zlib records only
    -> [(read_id, signal field bytes)]"""
    d = open(path, "rb").read()
    assert d[:6] == b"BLOW5\x01" and d[9] == 1 and d[14] == 1
    hs, = struct.unpack("<I", d[64:68])
    off = 68 + hs
    out = []
    while d[off:off + 5] != b"5WOLB":
        rs, = struct.unpack("<Q", d[off:off + 8])
        rec = zlib.decompress(d[off + 8:off + 8 + rs])
        off += 8 + rs
        idl, = struct.unpack("<H", rec[:2])
        p = 2 + idl + 4 + 32
        ln, = struct.unpack("<Q", rec[p:p + 8])
        out.append((rec[2:2 + idl].decode(), rec[p + 8:p + 8 + ln]))
    return out


def golden_reads():
    meta = json.load(open(os.path.join(GOLD, "three_reads.json")))
    sig = np.fromfile(os.path.join(GOLD, "three_reads.i16.bin"), dtype=np.int16)
    out, o = {}, 0
    for r in meta["reads"]:
        out[r["read_id"]] = sig[o:o + r["n"]]
        o += r["n"]
    return out


def test_reader_matches_the_file():
    """the C++ reader (no GPU involved) delivers the signal fields byte for byte, in file order,
    with the right ids and counts; small arenas split the file into several batches"""
    want = parse_blow5_py(BLOW5)
    for arena in (1 << 22, 200000, 330000):
        rd = press.Blow5Reader(BLOW5)
        assert (rd.record_method, rd.signal_method) == (1, 1)
        got = []
        while True:
            b = rd.next_batch(max_reads=2, arena_bytes=arena)
            if not b:
                break
            got += b
        rd.close()
        assert [(i, s) for i, _, s in got] == want
        for (_, n, s) in got:
            assert n == struct.unpack("<I", s[:4])[0]
    rd = press.Blow5Reader(BLOW5)
    with pytest.raises(press.PressError):
        rd.next_batch(max_reads=4, arena_bytes=1000)  # not even one signal fits
    rd.close()
    with pytest.raises(press.PressError):
        press.Blow5Reader(os.path.join(GOLD, "three_reads.i16.bin"))  # not a BLOW5 file


def test_oracle_on_the_files_own_streams(oracle):
    """oracle: decoding the file's svb-zd fields gives the samples the reference's loader gives,
    and encoding those samples gives the file's bytes back"""
    gold = golden_reads()
    for rid, stream in parse_blow5_py(BLOW5):
        sig = gold[rid]
        ret, dec = oracle.depress("slow5_svb_zd", stream, sig.size)
        assert ret == 0 and np.array_equal(dec, sig), rid
        ret, enc = oracle.press("slow5_svb_zd", sig)
        assert ret == 0 and enc == stream, rid
        # byte-compatible with the reference's svb_zd stream behind the 4-byte count (press.c:1590)
        ret, svb = oracle.press("svb_zd", sig)
        assert ret == 0 and stream[4:] == svb and struct.unpack("<I", stream[:4])[0] == sig.size


def test_oracle_malformed(oracle):
    sig = np.arange(100, dtype=np.int16)
    ret, s = oracle.press("slow5_svb_zd", sig)
    assert ret == 0
    for bad in (s[:3], s[:-1], s + b"\0", struct.pack("<I", 101) + s[4:]):
        ret, _ = oracle.depress("slow5_svb_zd", bad, 200)
        assert ret != 0
    ret, _ = oracle.depress("slow5_svb_zd", s, 99)  # no room
    assert ret != 0


@pytest.mark.gpu
def test_gpu_decodes_and_encodes_the_file(oracle):
    """device: file -> reader -> batch decode on the GPU == the reference loader's samples; the
    GPU's encoding of those samples == the file's own bytes"""
    press.load_library()
    gold = golden_reads()
    rd = press.Blow5Reader(BLOW5)
    batch = rd.next_batch()
    rd.close()
    assert len(batch) == 3
    back = press.depress_batch_host("slow5_svb_zd", [s for _, _, s in batch], [n for _, n, _ in batch])
    for (rid, n, _), b in zip(batch, back):
        assert b is not None and np.array_equal(b, gold[rid]), rid
    streams = press.press_batch_host("slow5_svb_zd", [gold[rid] for rid, _, _ in batch])
    for (rid, _, s), st in zip(batch, streams):
        assert st == s, rid
    # per-read symbols too
    for rid, n, s in batch:
        ret, dec = press.depress("slow5_svb_zd", s, n)
        assert ret == 0 and np.array_equal(dec, gold[rid])
        ret, enc = press.press("slow5_svb_zd", gold[rid])
        assert ret == 0 and enc == s


@pytest.mark.gpu
def test_gpu_wide_jumps_and_ragged(oracle):
    """jumps of more than 32767 (3-byte values, slow5lib's 32-bit delta), every tail length,
    multi-chunk reads; malformed streams fail like the oracle says"""
    press.load_library()
    rng = np.random.default_rng(5)
    cases = []
    for n in list(range(1, 20)) + [511, 512, 513, 4095, 32767, 32768, 32769, 70001]:
        kind = n % 3
        if kind == 0:
            s = rng.integers(-32768, 32768, size=n).astype(np.int16)  # most jumps are wide
        elif kind == 1:
            s = np.cumsum(rng.integers(-30, 31, size=n)).astype(np.int16)
            s[rng.integers(0, n, size=max(1, n // 500))] = rng.choice([-32768, 32767])
        else:
            s = np.cumsum(rng.integers(-300, 301, size=n)).astype(np.int16)
        cases.append(s)
    streams = press.press_batch_host("slow5_svb_zd", cases)
    for s, st in zip(cases, streams):
        ret, want = oracle.press("slow5_svb_zd", s)
        assert ret == 0 and st == want, s.size
    back = press.depress_batch_host("slow5_svb_zd", streams, [s.size for s in cases])
    for s, b in zip(cases, back):
        assert b is not None and np.array_equal(b, s), s.size
    good = streams[-1]
    n = cases[-1].size
    for bad in (good[:3], good[:-1], good + b"\0", struct.pack("<I", n + 1) + good[4:]):
        ro, _ = oracle.depress("slow5_svb_zd", bad, n + 8)
        rg, bg = press.depress("slow5_svb_zd", bad, n + 8)
        assert ro != 0 and rg != 0


def _ref_dump(path, tmp_path, ids=()):
    """the reference's slow5lib reads `path` (oracle/_ref/ref_dump_blow5) -> {read_id: samples}; with ids: those
    reads through the index (slow5_idx_load + slow5_get)"""
    import subprocess
    tool = os.path.join(_libs.ORACLE_DIR, "_ref", "ref_dump_blow5")
    if not os.path.exists(tool):
        pytest.skip("oracle/_ref/ref_dump_blow5 is not built here")
    out = str(tmp_path / "dump.bin")
    subprocess.run([tool, path, out] + list(ids), check=True, stderr=subprocess.DEVNULL)
    d = open(out, "rb").read()
    nreads, = struct.unpack("<I", d[:4])
    p, res = 4, {}
    for _ in range(nreads):
        idl, = struct.unpack("<I", d[p:p + 4])
        rid = d[p + 4:p + 4 + idl].decode()
        n, = struct.unpack("<Q", d[p + 4 + idl:p + 12 + idl])
        p += 12 + idl
        res[rid] = np.frombuffer(d[p:p + 2 * n], dtype=np.int16)
        p += 2 * n
    return res


@pytest.mark.gpu
@pytest.mark.parametrize("rec,sig", [(0, 1), (1, 1), (1, 0)])
def test_gpu_transcode_is_read_by_the_reference(tmp_path, rec, sig):
    """BLOW5 -> (reader, device svb-zd decode, device svb-zd encode, writer) -> BLOW5: the
    reference's own slow5lib opens the result and finds the same reads; with record compression
    off the svb-zd fields are the bytes of the source file"""
    press.load_library()
    dst = str(tmp_path / "out.blow5")
    assert press.blow5_transcode(BLOW5, dst, record_method=rec, signal_method=sig) == 3
    gold = golden_reads()
    got = _ref_dump(dst, tmp_path)
    assert set(got) == set(gold)
    for rid in gold:
        assert np.array_equal(got[rid], gold[rid]), rid
    rd = press.Blow5Reader(dst)
    assert (rd.record_method, rd.signal_method) == (rec, sig)
    if sig == 1:
        assert [(i, s) for i, _, s in rd.next_batch()] == parse_blow5_py(BLOW5)
    rd.close()


@pytest.mark.parametrize("rec", [0, 1])
def test_reframe_without_gpu(tmp_path, rec):
    """reader -> writer with the signal fields passed through (no GPU): the reference's slow5lib
    reads the re-framed file, and our reader gets the same fields back"""
    dst = str(tmp_path / "reframed.blow5")
    assert press.blow5_transcode(BLOW5, dst, record_method=rec, signal_method=1, passthrough=True) == 3
    rd = press.Blow5Reader(dst)
    assert (rd.record_method, rd.signal_method) == (rec, 1)
    assert [(i, s) for i, _, s in rd.next_batch()] == parse_blow5_py(BLOW5)
    rd.close()
    got = _ref_dump(dst, tmp_path)
    gold = golden_reads()
    assert set(got) == set(gold) and all(np.array_equal(got[k], gold[k]) for k in gold)


@pytest.mark.parametrize("rec", [0, 1])
def test_index_is_the_one_slow5lib_writes(tmp_path, rec):
    """SURVEY 8f-2 / slow5_idx.c: the writer's <file>.idx lets the reference's slow5_idx_load / slow5_get find reads of
    a file this library wrote - and it is byte for byte the index slow5lib builds for that file itself"""
    fields = [f for _, f in parse_blow5_py(BLOW5)]
    gold = list(golden_reads().values())
    # 40 records (the three real signals over and over) under ids of their own
    many = [fields[k % 3] for k in range(40)]
    dst = str(tmp_path / "indexed.blow5")
    ids = press.blow5_write_like(dst, BLOW5, many, record_method=rec, signal_method=1, index=True)
    assert len(ids) == 40 and os.path.exists(dst + ".idx")
    ours = open(dst + ".idx", "rb").read()
    assert ours[:9] == b"SLOW5IDX\x01" and ours[-8:] == b"XDI5WOLS"
    # random access by id, through OUR index
    pick = [ids[37], ids[0], ids[20], ids[5]]
    got = _ref_dump(dst, tmp_path, pick)
    assert list(got) == pick
    for rid in pick:
        assert np.array_equal(got[rid], gold[int(rid.split("-")[1]) % 3]), rid
    assert open(dst + ".idx", "rb").read() == ours  # (slow5lib read it, it did not rebuild it)
    # ... and the index slow5lib builds when there is none is the same bytes
    os.remove(dst + ".idx")
    _ref_dump(dst, tmp_path, pick[:1])
    assert open(dst + ".idx", "rb").read() == ours
    # the transcoder writes one too
    dst2 = str(tmp_path / "t.blow5")
    assert press.blow5_transcode(BLOW5, dst2, record_method=rec, signal_method=1, passthrough=True, index=True) == 3
    got = _ref_dump(dst2, tmp_path, [i for i, _ in parse_blow5_py(BLOW5)][::-1])
    assert len(got) == 3 and all(np.array_equal(got[k], golden_reads()[k]) for k in got)


def test_reader_thread_pool_keeps_the_order(tmp_path):
    """records are inflated by a pool of host threads: the batches come out in file order whatever the pool size and
    the batch limits, byte for byte"""
    fields = [f for _, f in parse_blow5_py(BLOW5)]
    many = [fields[k % 3][: 4 + (len(fields[k % 3]) - 4) // (1 + k % 5)] for k in range(90)]  # (fields of many sizes)
    # (truncated svb-zd fields are fine for the reader: it hands fields out as stored)
    dst = str(tmp_path / "many.blow5")
    ids = press.blow5_write_like(dst, BLOW5, many, record_method=1, signal_method=1)
    for threads, maxr, arena in ((1, 7, 1 << 24), (4, 64, 1 << 24), (0, 4096, 1 << 24), (3, 50, 600000)):
        rd = press.Blow5Reader(dst)
        rd.set_threads(threads)
        got = []
        while True:
            b = rd.next_batch(max_reads=maxr, arena_bytes=arena)
            if not b:
                break
            got += b
        rd.close()
        assert [g[0] for g in got] == ids
        assert [g[2] for g in got] == [bytes(m) for m in many]

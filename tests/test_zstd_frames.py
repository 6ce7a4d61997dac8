"""zstd frames made on the device (batch method zstd_svb_zd).

What pins them: any zstd decoder must turn a frame into the buffer the reference gives to
ZSTD_compress (press.c:1860: [u32 n][svb-zd stream]).  The compressed BYTES are free - the
reference's own depend on the libzstd version ("parity unpinned").

* CPU: oracle/zsframe_model.cpp (a serial model of the device writer, built on the product's
  table code zs_table.h) -> libzstd must decode every frame; ratio against libzstd level 1.
* GPU: the device writes the model's bytes; libzstd and the oracle's zstd_svb_zd decoder (the
  reference's, when built) read them; the device reads them back.
"""
import ctypes
import os
import struct
import subprocess

import numpy as np
import pytest

import _libs
from honours_amd import synth

ROOT = _libs.ROOT
MODEL_SRC = os.path.join(ROOT, "oracle", "zsframe_model.cpp")
MODEL_SO = os.path.join(ROOT, "oracle", "libzsframe_model.so")


def _zstd():
    from honours_amd import press
    z = press.open_libzstd()
    if z is None:
        pytest.skip("no libzstd")
    return z


def zstd_decode(z, frame, cap):
    a = np.frombuffer(frame, dtype=np.uint8).copy()
    out = np.zeros(cap + 64, dtype=np.uint8)
    r = z.ZSTD_decompress(out.ctypes.data, cap + 64, a.ctypes.data, len(frame))
    assert not z.ZSTD_isError(r), "libzstd refuses the frame"
    return out[:r].tobytes()


def zstd_level1(z, buf):
    a = np.frombuffer(buf, dtype=np.uint8).copy()
    out = np.zeros(len(buf) + len(buf) // 100 + 1024, dtype=np.uint8)
    r = z.ZSTD_compress(out.ctypes.data, out.size, a.ctypes.data, len(buf), 1)
    assert not z.ZSTD_isError(r)
    return int(r)


@pytest.fixture(scope="module")
def model():
    if not os.path.exists(MODEL_SO) or os.path.getmtime(MODEL_SO) < max(
            os.path.getmtime(MODEL_SRC), os.path.getmtime(os.path.join(ROOT, "honours_amd", "csrc", "zs_table.h"))):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "zsmodel"], check=True)
    m = ctypes.CDLL(MODEL_SO)
    m.zsm_frame_k.restype = ctypes.c_uint64
    m.zsm_frame_k.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32]

    m.zsm_frame_p.restype = ctypes.c_uint64
    m.zsm_frame_p.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64,
                              ctypes.c_uint64]

    def frame(buf, kdiv=4):
        a = np.frombuffer(buf, dtype=np.uint8).copy()
        out = np.zeros(len(buf) + len(buf) // 100 + 65536, dtype=np.uint8)
        if kdiv:
            r = m.zsm_frame_k(a.ctypes.data, len(buf), out.ctypes.data, out.size, kdiv)
        else:
            # ex-zd (ex_zd.c:9): version, u64 n, q, u16 zd0, u32 nex, exception section, then one byte for
            # every other sample: the prefix is what is in front of those bytes
            n, = struct.unpack("<Q", buf[1:9])
            nex, = struct.unpack("<I", buf[12:16])
            r = m.zsm_frame_p(a.ctypes.data, len(buf), out.ctypes.data, out.size, len(buf) - (n - 1 - nex), 0)
        assert r
        return out[:r].tobytes()
    return frame


# batch method -> (inner stream, samples per key byte)
KINDS = {"zstd_svb_zd": ("svb_zd", 4), "zstd_svb12_zd": ("svb12_zd", 8),
         "zstd_hasgam_vbsse21_zdq": ("hasgam_vbsse21_zdq", 0)}


def prezstd(oracle, s, inner="svb_zd"):
    """the buffer the reference hands to ZSTD_compress (press.c:1860, 2020, 8554)"""
    ret, c = oracle.press(inner, s)
    if ret != 0 and inner.startswith("hasgam"):
        return None  # more exceptions than the reference's bound has room for (press.c:2575): no such stream
    assert ret == 0
    return c if inner.startswith("hasgam") else struct.pack("<I", len(s)) + c


def cases():
    rng = np.random.default_rng(11)
    out = []
    sig, off = synth.synth_batch(7, 0, 10)
    for r in range(10):
        out.append(sig[int(off[r]):int(off[r + 1])])
    for n in (1, 2, 3, 4, 5, 63, 64, 65, 100, 1000, 4096, 16384 + 5, 70000, 140000):
        out.append(rng.integers(300, 700, n).astype(np.int16))                    # noisy
        out.append(np.full(n, 511, dtype=np.int16))                               # constant: one data byte value
        out.append(rng.integers(-32768, 32767, n).astype(np.int16))               # incompressible, every key set
        out.append((np.cumsum(rng.integers(-3, 4, n)) + 500).astype(np.int16))    # tiny alphabet
        s = (np.cumsum(rng.integers(-20, 21, n)) + 500).astype(np.int16)          # spikes: scattered key bytes
        s[rng.integers(0, n, max(1, n // 300))] += 3000
        out.append(s)
    return out


@pytest.mark.parametrize("zm", sorted(KINDS))
def test_model_frames_decode_with_libzstd(model, zm):
    z = _zstd()
    oracle = _libs.oracle()
    inner, kdiv = KINDS[zm]
    raw = ours = theirs = 0
    for k, s in enumerate(cases()):
        buf = prezstd(oracle, s, inner)
        if buf is None:
            continue
        f = model(buf, kdiv)
        assert zstd_decode(z, f, len(buf)) == buf, "case %d (n=%d)" % (k, len(s))
        assert len(f) <= 9 + len(buf) + 3 * ((len(buf) + 131071) // 131072)
        if k < 10:  # the NA12878-like reads
            raw += 2 * len(s)
            ours += len(f)
            theirs += zstd_level1(z, buf)
    # within 0.5 % of libzstd level 1 (what the reference calls VBZ) on nanopore-like reads
    assert raw / ours > 0.995 * raw / theirs, (raw / ours, raw / theirs)


def test_table_is_a_complete_prefix_code(model):
    """zs_table.h on random histograms: Kraft equality, lengths <= 11, canonical codes"""
    m = ctypes.CDLL(MODEL_SO)

    class T(ctypes.Structure):
        _fields_ = [("code", ctypes.c_uint16 * 256), ("len", ctypes.c_uint8 * 256), ("desc", ctypes.c_uint8 * 132),
                    ("desc_len", ctypes.c_uint32), ("table_log", ctypes.c_uint32), ("ok", ctypes.c_uint32),
                    ("pad", ctypes.c_uint32)]
    m.zsm_table.argtypes = [ctypes.c_void_p, ctypes.POINTER(T)]
    rng = np.random.default_rng(3)
    for trial in range(300):
        k = int(rng.integers(2, 257))
        cnt = np.zeros(256, dtype=np.uint32)
        syms = rng.choice(256, k, replace=False)
        kind = trial % 4
        if kind == 0:
            cnt[syms] = rng.integers(1, 1000, k)
        elif kind == 1:
            cnt[syms] = np.maximum(1, (1e6 * rng.random(k) ** 8).astype(np.uint32))   # steep
        elif kind == 2:
            cnt[syms] = 1 + (np.arange(k) == 0) * 10 ** 6                              # one giant
        else:
            fib = [1, 1]
            while len(fib) < k:
                fib.append(min(fib[-1] + fib[-2], 2 ** 31 // 300))
            cnt[syms] = fib[:k]                                                       # deepest possible tree
        t = T()
        m.zsm_table(cnt.ctypes.data, ctypes.byref(t))
        ln = np.array(list(t.len), dtype=np.int64)
        if not t.ok:
            # only a flat 256-byte alphabet has no description (and does not shrink anyway)
            assert k > 128 and len(set(ln[ln > 0])) <= 1
            continue
        assert ((ln > 0) == (cnt > 0)).all() and ln.max() <= 11 and ln.max() == t.table_log
        assert sum(2.0 ** -l for l in ln if l) == 1.0
        codes = sorted((int(ln[s]), int(t.code[s])) for s in range(256) if ln[s])
        seen = set()
        for l, c in codes:  # prefix free
            assert c < (1 << l)
            for p in range(1, l):
                assert (p, c >> (l - p)) not in seen
            seen.add((l, c))


def test_reader_survives_damaged_frames():
    """the device's frame reader (zs_table.h) compiled for the host with AddressSanitizer + UBSan on
    10 000 truncated / extended / bit-flipped frames: stays inside the frame and inside its output"""
    exe = os.path.join(ROOT, "oracle", "zsframe_fuzz")
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "zsfuzz"], check=True)
    r = subprocess.run([exe, "10000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "no finding" in r.stdout


# ------------------------------------------------------------------ GPU

gpu = pytest.mark.gpu


@gpu
@pytest.mark.parametrize("zm", sorted(KINDS))
def test_device_frames_match_the_model_and_decode(model, zm):
    from honours_amd import press
    z = _zstd()
    oracle = _libs.oracle()
    inner, kdiv = KINDS[zm]
    reads = cases()
    frames = press.press_batch_host(zm, reads)
    for k, (s, f) in enumerate(zip(reads, frames)):
        buf = prezstd(oracle, s, inner)
        if buf is None:
            continue
        assert f is not None, "case %d" % k
        assert zstd_decode(z, f, len(buf)) == buf, "case %d (n=%d): libzstd" % (k, len(s))
        assert f == model(buf, kdiv), "case %d (n=%d): not the model's bytes" % (k, len(s))
        if len(buf) > 2 * len(s):
            continue  # press.c:1896: the reference's decoder sizes its buffer for 2 bytes per sample (+ zstd's margin)
        ret, back = oracle.depress(zm, f, s.size)   # the oracle's decoder (libzstd + the inner stream's)
        assert ret == 0 and np.array_equal(back, s)
        if _libs.have_reference():
            ret, back = _libs.reference().depress(zm, f, s.size)
            assert ret == 0 and np.array_equal(back, s)


@gpu
def test_device_frames_small_slots():
    """a slot that is too small fails that read only"""
    from honours_amd import press
    rng = np.random.default_rng(5)
    reads = [rng.integers(300, 700, 50000).astype(np.int16) for _ in range(3)]
    full = press.press_batch_host("zstd_svb_zd", reads)
    caps = [len(full[0]) + 64, len(full[1]) - 32, len(full[2]) + 64]  # slots are rounded up to 16 bytes
    got = press.press_batch_host("zstd_svb_zd", reads, caps=caps)
    assert got[0] == full[0] and got[1] is None and got[2] == full[2]


def test_model_reader_on_both_kinds_of_frames(model):
    """zs::walk_frame on the host (the code the device walks frames with): this library's frames decode, and
    so do libzstd's own at the reference's level 1 (press.h:275) and at levels 3 and 9 - sequences
    included (FSE tables of all four modes, repeat offsets, overlapping matches)"""
    z = _zstd()
    oracle = _libs.oracle()
    m = ctypes.CDLL(MODEL_SO)
    m.zsm_decode.restype = ctypes.c_int64
    m.zsm_decode.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64]

    def dec(f, cap):
        a = np.frombuffer(f, dtype=np.uint8).copy()
        out = np.zeros(cap + 64, dtype=np.uint8)
        r = m.zsm_decode(a.ctypes.data, len(f), out.ctypes.data, cap)
        return r, out[:max(r, 0)].tobytes()
    with_seq = 0
    for s in cases():
        buf = prezstd(oracle, s)
        r, b = dec(model(buf), len(buf))
        assert r == len(buf) and b == buf
        a = np.frombuffer(buf, dtype=np.uint8).copy()
        outz = np.zeros(len(buf) + len(buf) // 100 + 1024, dtype=np.uint8)
        for level in (1, 3, 9):
            rz = z.ZSTD_compress(outz.ctypes.data, outz.size, a.ctypes.data, len(buf), level)
            r, b = dec(outz[:rz].tobytes(), len(buf))
            assert r == len(buf) and b == buf, (len(s), level, r)
        with_seq += 1
    # byte streams with long and overlapping matches, short offsets, RLE tables
    rng = np.random.default_rng(5)
    texts = [bytes(rng.integers(0, 4, 70000, dtype=np.uint8)), b"abcabcabd" * 9000, bytes(200000),
             bytes(rng.integers(0, 256, 300, dtype=np.uint8)) * 700, b"x" * 5 + bytes(range(256)) * 600]
    for t in texts:
        a = np.frombuffer(t, dtype=np.uint8).copy()
        outz = np.zeros(len(t) + len(t) // 100 + 1024, dtype=np.uint8)
        for level in (1, 3, 9, 19):
            rz = z.ZSTD_compress(outz.ctypes.data, outz.size, a.ctypes.data, len(t), level)
            r, b = dec(outz[:rz].tobytes(), len(t))
            assert r == len(t) and b == t, (len(t), level, r)
    assert with_seq > 10


def _libzstd_frames(z, oracle, reads, level=1, inner="svb_zd"):
    out = []
    for s in reads:
        buf = prezstd(oracle, s, inner)
        if buf is None:
            out.append(None)
            continue
        a = np.frombuffer(buf, dtype=np.uint8).copy()
        o = np.zeros(len(buf) + len(buf) // 100 + 1024, dtype=np.uint8)
        r = z.ZSTD_compress(o.ctypes.data, o.size, a.ctypes.data, len(buf), level)
        out.append(o[:r].tobytes())
    return out


@gpu
@pytest.mark.parametrize("zm", sorted(KINDS))
def test_device_reads_its_own_frames(zm):
    from honours_amd import press
    reads = cases()
    frames = press.press_batch_host(zm, reads)
    if zm.startswith("zstd_hasgam"):  # reads with more exceptions than the format's bound allows have no stream
        reads = [s for s, f in zip(reads, frames) if f is not None]
        frames = [f for f in frames if f is not None]
        assert len(reads) >= 60
    back = press.depress_batch_host(zm, frames, [len(s) for s in reads])
    for k, (s, b) in enumerate(zip(reads, back)):
        assert b is not None and np.array_equal(b, s), "case %d (n=%d)" % (k, len(s))
    # a larger slot than the read: the count in the stream decides (press.c:1901)
    back = press.depress_batch_host(zm, frames, [len(s) + 1000 for s in reads])
    for s, b in zip(reads, back):
        assert b is not None and np.array_equal(b, s)


@gpu
@pytest.mark.parametrize("zm", sorted(KINDS))
def test_device_reads_libzstd_frames(zm):
    """the reference's own streams (ZSTD_compress level 1, press.h:275; levels 3 and 9 as well): decoded
    entirely on the device - sequences included, no frame goes to libzstd on the host"""
    from honours_amd import press
    z = _zstd()
    oracle = _libs.oracle()
    reads = cases()
    lib = press.load_library()
    for level in (1, 3, 9):
        frames = _libzstd_frames(z, oracle, reads, level, KINDS[zm][0])
        keep = [k for k, f in enumerate(frames) if f is not None]
        back = press.depress_batch_host(zm, [frames[k] for k in keep], [len(reads[k]) for k in keep])
        assert lib.press_hip_zstd_host_frames() == 0, level
        for k, b in zip(keep, back):
            assert b is not None and np.array_equal(b, reads[k]), "level %d case %d (n=%d)" % (level, k, len(reads[k]))
    # and the per-read symbol of the reference's interface reads a device-made frame
    f = press.press_batch_host(zm, reads[:2])
    for s, fr in zip(reads[:2], f):
        ret, b = press.depress(zm, fr, len(s))
        assert ret == 0 and np.array_equal(b, s)


@gpu
def test_long_streams_of_libzstd_frames():
    """ZSTD_compress puts up to 128 KiB of literals in a block: four streams of up to 32 768 codes, which the device
    decodes with 16 lanes each (k_zs_hdecode_long: segments that synchronise themselves).  Reads of several such
    blocks, noisy / tiny alphabet / a real-looking read, levels 1 and 3; then the same frames damaged inside a long stream."""
    from honours_amd import press
    z = _zstd()
    oracle = _libs.oracle()
    lib = press.load_library()
    rng = np.random.default_rng(23)
    reads = [rng.integers(300, 700, 600000).astype(np.int16),                      # ~5 blocks, nearly no matches
             (np.cumsum(rng.integers(-3, 4, 300001)) + 500).astype(np.int16),     # tiny alphabet: short codes
             rng.integers(480, 520, 131072 + 7).astype(np.int16)]
    n, first = synth.read_lengths(3, 0, 64)
    k = int(np.argmax(n))
    reads.append(synth.synth_read(3, k, int(n[k]), int(first[k])))
    # sixteen equally likely byte values: every code has 4 bits, and a decoder that starts between two codes NEVER falls
    # into step - the segments are then put right one per round, from the stream's true start down (the worst case of
    # the scheme: as slow as one lane, and as right)
    reads.append((np.cumsum(rng.integers(-8, 8, 400000)) + 500).astype(np.int16))
    reads.append((np.cumsum(rng.integers(-2, 2, 300000)) + 500).astype(np.int16))  # four values: 2-bit codes
    for level in (1, 3):
        frames = _libzstd_frames(z, oracle, reads, level)
        back = press.depress_batch_host("zstd_svb_zd", frames, [len(s) for s in reads])
        assert lib.press_hip_zstd_host_frames() == 0, level
        for s, b in zip(reads, back):
            assert b is not None and np.array_equal(b, s), "level %d n=%d" % (level, len(s))
    frames = _libzstd_frames(z, oracle, reads, 1)
    bad = list(frames)
    for i in (0, 1):
        x = bytearray(frames[i])
        for at in (len(x) // 3, len(x) // 2):
            x[at] ^= 0x24
        bad[i] = bytes(x)
    bad[2] = frames[2][:len(frames[2]) - 9]
    back = press.depress_batch_host("zstd_svb_zd", bad, [len(s) for s in reads])
    for i in range(3, len(reads)):
        assert np.array_equal(back[i], reads[i])
    assert back[2] is None
    for i in (0, 1):  # a flipped bit breaks a stream's end or its code count (refused) or decodes to other samples
        assert back[i] is None or not np.array_equal(back[i], reads[i])


@gpu
def test_device_refuses_damaged_frames():
    from honours_amd import press
    rng = np.random.default_rng(9)
    reads = [rng.integers(300, 700, 40000).astype(np.int16) for _ in range(6)]
    good = press.press_batch_host("zstd_svb_zd", reads)
    bad = list(good)
    bad[1] = good[1][:len(good[1]) // 2]                      # truncated
    bad[2] = b"\x00" + good[2][1:]                            # no magic
    x = bytearray(good[3])
    x[len(x) // 2] ^= 0x55                                    # a flipped byte inside a Huffman stream
    bad[3] = bytes(x)
    bad[4] = good[4] + b"\x00\x00\x00"                        # trailing bytes
    back = press.depress_batch_host("zstd_svb_zd", bad, [len(s) for s in reads])
    assert np.array_equal(back[0], reads[0]) and np.array_equal(back[5], reads[5])
    assert back[1] is None and back[2] is None and back[4] is None
    # a flipped bit either breaks a stream's end (refused) or decodes to other samples - never the read
    assert back[3] is None or not np.array_equal(back[3], reads[3])
    # too small a slot for the count in the stream
    back = press.depress_batch_host("zstd_svb_zd", good[:2], [len(reads[0]), len(reads[1]) - 1])
    assert np.array_equal(back[0], reads[0]) and back[1] is None


@gpu
def test_device_frames_of_empty_and_tiny_reads():
    """a batch that mixes an empty read, one-sample reads and a normal one (svb kinds: an empty read is a
    frame of the bare count, as ZSTD_compress of press.c:1865 would see it)"""
    from honours_amd import press
    z = _zstd()
    rng = np.random.default_rng(21)
    reads = [np.zeros(0, dtype=np.int16), np.array([500], dtype=np.int16), rng.integers(300, 700, 30000).astype(np.int16),
             np.array([-32768], dtype=np.int16), np.zeros(0, dtype=np.int16)]
    for zm in ("zstd_svb_zd", "zstd_svb12_zd"):
        frames = press.press_batch_host(zm, reads)
        for s, f in zip(reads, frames):
            assert f is not None
            got = zstd_decode(z, f, 4 + 3 * len(s) + 16)
            assert struct.unpack("<I", got[:4])[0] == len(s)
        back = press.depress_batch_host(zm, frames, [max(1, len(s)) for s in reads])
        for s, b in zip(reads, back):
            assert b is not None and np.array_equal(b, s)


@gpu
@pytest.mark.parametrize("zm", sorted(KINDS))
def test_three_reads_fixture(zm):
    """config 1 of BASELINE.json (data/three-reads.blow5, VBZ): the device's frames of the reference's own
    reads are read by libzstd + the oracle, and are within 1 % of the size the reference got from libzstd
    (tests/golden/three_reads.json: zstd_svb_zd 176598 bytes in all)"""
    import json
    from honours_amd import press
    oracle = _libs.oracle()
    gdir = os.path.join(ROOT, "tests", "golden")
    meta = json.load(open(os.path.join(gdir, "three_reads.json")))
    sig = np.fromfile(os.path.join(gdir, "three_reads.i16.bin"), dtype=np.int16)
    reads, o = [], 0
    for r in meta["reads"]:
        reads.append(sig[o:o + r["n"]])
        o += r["n"]
    frames = press.press_batch_host(zm, reads)
    ours = ref = 0
    for r, s, f in zip(meta["reads"], reads, frames):
        assert f is not None
        ret, back = oracle.depress(zm, f, s.size)
        assert ret == 0 and np.array_equal(back, s)
        ours += len(f)
        ref += r["methods"][zm]["len"]
    assert ours <= ref * 1.01, (ours, ref)
    back = press.depress_batch_host(zm, frames, [len(s) for s in reads])
    for s, b in zip(reads, back):
        assert np.array_equal(b, s)
    print("%s on three-reads.blow5: %d bytes (the reference's libzstd: %d)" % (zm, ours, ref))


@gpu
def test_device_reader_on_many_damaged_frames():
    """300 truncated / bit-flipped frames in one batch: whatever the host build of the same reader
    (oracle/zsframe_model.cpp: zs::walk_frame) refuses, the device refuses; undamaged neighbours decode"""
    from honours_amd import press
    m = ctypes.CDLL(MODEL_SO)
    m.zsm_decode.restype = ctypes.c_int64
    m.zsm_decode.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64]
    rng = np.random.default_rng(77)
    base = [rng.integers(300, 700, int(n)).astype(np.int16) for n in (3000, 20000, 70000)]
    good = press.press_batch_host("zstd_svb_zd", base)
    reads, frames, verdict = [], [], []
    for k in range(300):
        j = k % 3
        f = bytearray(good[j])
        kind = k % 5
        if kind == 0:
            f = f[:int(rng.integers(0, len(f)))]
        elif kind == 4:
            pass  # undamaged
        else:
            for _ in range(int(rng.integers(1, 4))):
                at = int(rng.integers(0, min(len(f), 64))) if rng.integers(0, 3) == 0 else int(rng.integers(0, len(f)))
                f[at] ^= 1 << int(rng.integers(0, 8))
        f = bytes(f) if len(f) else b"\0"
        a = np.frombuffer(f, dtype=np.uint8).copy()
        cap = 4 + (len(base[j]) + 3) // 4 + 2 * len(base[j])
        out = np.zeros(cap + 64, dtype=np.uint8)
        verdict.append(int(m.zsm_decode(a.ctypes.data, len(f), out.ctypes.data, cap)))
        reads.append(base[j])
        frames.append(f)
    back = press.depress_batch_host("zstd_svb_zd", frames, [len(s) for s in reads])
    refused = 0
    for k, (s, b, v) in enumerate(zip(reads, back, verdict)):
        if k % 5 == 4:
            assert b is not None and np.array_equal(b, s), k
        if v == -1:
            assert b is None, "frame %d: the host reader refuses it, the device does not" % k
            refused += 1
    assert refused > 50


FORGED_LITERALS = bytes([0x28, 0xB5, 0x2F, 0xFD, 0x20, 24, 0x4D, 0x00, 0x00,   # frame header, one last compressed block of 9 bytes
                         0x0D, 0x00, 0x20, 0xAA,                                # literals: RLE, R = 131072
                         0x01, 0x00, 0x01, 0x00, 0x80])                         # one sequence, predefined tables


def test_model_refuses_forged_literals():
    """a block with sequences whose literals do not fit the room is refused before anything is queued"""
    m = ctypes.CDLL(MODEL_SO)
    m.zsm_decode.restype = ctypes.c_int64
    m.zsm_decode.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64]
    a = np.frombuffer(FORGED_LITERALS, dtype=np.uint8).copy()
    out = np.zeros(64, dtype=np.uint8)
    assert m.zsm_decode(a.ctypes.data, a.size, out.ctypes.data, 27) == -1
    assert not out.any()


@gpu
def test_device_refuses_forged_literals():
    """a ~18-byte frame whose RLE literals claim 131072 bytes, in the batch's LAST and shortest read
    (its literals slot is the end of the scratch arena): refused, the neighbours decode, the library reports no fault"""
    from honours_amd import press
    rng = np.random.default_rng(5)
    reads = [rng.integers(300, 700, int(n)).astype(np.int16) for n in (40000, 9000, 10)]
    good = press.press_batch_host("zstd_svb_zd", reads)
    for at in (2, 1, 0):
        bad = list(good)
        bad[at] = FORGED_LITERALS
        back = press.depress_batch_host("zstd_svb_zd", bad, [len(s) for s in reads])
        for k in range(3):
            if k == at:
                assert back[k] is None
            else:
                assert back[k] is not None and np.array_equal(back[k], reads[k]), (at, k)
    # and the undamaged batch still decodes afterwards
    back = press.depress_batch_host("zstd_svb_zd", good, [len(s) for s in reads])
    assert all(np.array_equal(b, s) for b, s in zip(back, reads))
    assert press.load_library().press_hip_synchronize() == 0


@gpu
def test_device_reader_on_damaged_libzstd_frames():
    """bit-flipped / truncated frames of ZSTD_compress at levels 1, 3 and 9 (sequences, repeated tables): what the
    host build of the reader refuses the device refuses, what it decodes the device decodes to the same samples or
    refuses later (the svb layer); undamaged neighbours are intact"""
    from honours_amd import press
    z = _zstd()
    oracle = _libs.oracle()
    m = ctypes.CDLL(MODEL_SO)
    m.zsm_decode.restype = ctypes.c_int64
    m.zsm_decode.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64]
    rng = np.random.default_rng(78)
    # reads with repeats, so that libzstd finds matches
    base = []
    for n in (3000, 20000, 70000):
        s = rng.integers(300, 700, int(n)).astype(np.int16)
        s[n // 2:n // 2 + n // 4] = s[:n // 4]
        base.append(s)
    reads, frames, verdict = [], [], []
    for level in (1, 3, 9):
        good = _libzstd_frames(z, oracle, base, level)
        for k in range(60):
            j = k % 3
            f = bytearray(good[j])
            kind = k % 5
            if kind == 0:
                f = f[:int(rng.integers(0, len(f)))]
            elif kind != 4:
                for _ in range(int(rng.integers(1, 4))):
                    at = int(rng.integers(0, min(len(f), 64))) if rng.integers(0, 3) == 0 else int(rng.integers(0, len(f)))
                    f[at] ^= 1 << int(rng.integers(0, 8))
            f = bytes(f) if len(f) else b"\0"
            a = np.frombuffer(f, dtype=np.uint8).copy()
            cap = 4 + (len(base[j]) + 3) // 4 + 2 * len(base[j])
            out = np.zeros(cap + 64, dtype=np.uint8)
            verdict.append(int(m.zsm_decode(a.ctypes.data, len(f), out.ctypes.data, cap)))
            reads.append(base[j])
            frames.append(f)
    back = press.depress_batch_host("zstd_svb_zd", frames, [len(s) for s in reads])
    refused = 0
    for k, (s, b, v) in enumerate(zip(reads, back, verdict)):
        if k % 5 == 4:
            assert b is not None and np.array_equal(b, s), k
        if v == -1:
            assert b is None, "frame %d: the host reader refuses it, the device does not" % k
            refused += 1
    assert refused > 20
    assert press.load_library().press_hip_synchronize() == 0

#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference.

Run in the build container only (needs /root/reference and `make -C oracle ref`):

    python tests/golden/make_golden.py

Reads the reference at fixture-build time only; commits DATA (inputs + expected
outputs), never reference code:

* three_reads.i16.bin / three_reads.json - the raw signals of the reference's own
  fixture data/three-reads.blow5 (decoded with the reference's slow5lib via
  oracle/_ref/ref_dump_blow5) and, per method, the compressed length + FNV-1a-32 +
  sha256 of the stream the reference produces (README:43-54 runs press/test on it);
* micro_kats.json - tiny inputs with the complete expected stream per method
  (edge cases: n = 1, 2, n%8 in {0,1,7}, 0/1/2/many exceptions, int16 wraparound,
  q = 0..5, zd[0] > 255);
* svb32_issue42.bin - the known-answer stream of press/streamvbyte/tests/unit.c:283
  (36 494 compressed bytes <-> 29 159 values);
* synth_kats.json - synthetic NA12878-like reads (honours_amd.synth, seeds only) with
  expected length + sha256 per method.
"""
import hashlib
import json
import os
import re
import struct
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)

import _libs  # noqa: E402
from honours_amd import synth  # noqa: E402

REF = "/root/reference"
# The range-coder methods store a few dozen samples raw without telling their decoder (TurboRC
# rcutil_.h:161) and the reference aborts on some of the tiny inputs below (wrap8 with rcc_vbe21_zd):
# reads that short are outside these methods' domain - their known answers are three_reads.json and the
# synthetic reads of 64 samples and more.
RCFAM = ("rc_", "rcc_", "rccm_")


def quiet_stderr():
    """the reference prints diagnostics on the hot path (press.c:3262) - drop them."""
    devnull = os.open(os.devnull, os.O_WRONLY)
    os.dup2(devnull, 2)


def sha(b):
    return hashlib.sha256(b).hexdigest()[:32]


def zd_of(sig):
    s = np.asarray(sig, dtype=np.int16).astype(np.uint16)
    d = (s - np.concatenate([[0], s[:-1]]).astype(np.uint16)).astype(np.int16)
    return ((d.astype(np.int32) << 1) ^ (d.astype(np.int32) >> 15)).astype(np.uint16)


def in_valid_domain(method, sig):
    """SURVEY 'Reference quirks' 2: the static-Huffman decoders need >= 1 one-byte symbol."""
    if method.startswith("shuffman"):
        z = zd_of(sig)
        return int((z[1:] <= 255).sum()) >= 1
    return True


def main():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True)
    ref = _libs.reference()
    quiet_stderr()

    # ---- 1. three-reads.blow5 --------------------------------------------------
    tmp = "/tmp/three_reads_dump.bin"
    subprocess.run([os.path.join(ROOT, "oracle/_ref/ref_dump_blow5"),
                    os.path.join(REF, "data/three-reads.blow5"), tmp], check=True)
    d = open(tmp, "rb").read()
    nr = struct.unpack_from("<I", d, 0)[0]
    o = 4
    reads, sigs = [], []
    for _ in range(nr):
        il = struct.unpack_from("<I", d, o)[0]; o += 4
        rid = d[o:o + il].decode(); o += il
        n = struct.unpack_from("<Q", d, o)[0]; o += 8
        sig = np.frombuffer(d, dtype=np.int16, count=n, offset=o).copy(); o += 2 * n
        ent = {"read_id": rid, "n": int(n), "methods": {}}
        for m in _libs.METHODS:
            ret, c = ref.press(m, sig)
            assert ret == 0, (m, ret)
            r2, back = ref.depress(m, c, n)
            assert r2 == 0 and np.array_equal(back, sig), m
            e = {"len": len(c), "bound": int(ref.bound(m, n))}
            if m in _libs.DETERMINISTIC:
                e["fnv1a32"] = "%08x" % _libs.fnv1a32(c)
                e["sha256_32"] = sha(c)
            ent["methods"][m] = e
        reads.append(ent)
        sigs.append(sig)
    np.concatenate(sigs).tofile(os.path.join(HERE, "three_reads.i16.bin"))
    json.dump({"source": "data/three-reads.blow5 decoded by the reference's slow5lib; streams by the "
                         "reference compiled from /root/reference/press (zstd 1.4.9 for zstd_* lengths)",
               "reads": reads}, open(os.path.join(HERE, "three_reads.json"), "w"), indent=1)

    # ---- 2. micro known answers ---------------------------------------------------
    micro_inputs = {
        "one": [500],
        "one_small": [17],
        "two": [500, 503],
        "five_one_exc": [100, 101, 400, 402, 401],
        "nine_two_exc": [500, 503, 499, 499, 900, 901, 300, 300, 305],
        "wrap8": [32767, -32768, -1, 0, 32767, -32767, 5, 6],
        "q5": [512, 544, 480, 480, 1056, 1056, 96, 128],
        "q1": [512, 514, 480, 486, 1056, 1058, 96, 130, 2],
        "q3": [8, 16, 24, 808, 816, 8, 0, -8, -16],
        "zeros9": [0] * 9,
        "seven": [600, 601, 603, 602, 604, 1000, 1001],
        "eight": [600, 601, 603, 602, 604, 1000, 1001, 1002],
        "sixteen_tail_exc": [600 + (i % 3) for i in range(15)] + [2000],
        "seventeen": [600 + (i * 7) % 5 for i in range(17)],
        "all_exc_10": [0, 1000, 0, 1000, 0, 1000, 0, 1000, 0, 1000],
        "many_exc_40": [((i * 37) % 11) * (300 if i % 3 == 0 else 1) for i in range(40)],
        "exc_far_apart_300": [500 + (i % 7) for i in range(300)],
    }
    micro_inputs["exc_far_apart_300"][5] = 1500
    micro_inputs["exc_far_apart_300"][290] = -1200
    micro = []
    for name, vals in micro_inputs.items():
        sig = np.array(vals, dtype=np.int16)
        ent = {"name": name, "input": [int(v) for v in sig], "streams": {}}
        for m in _libs.DETERMINISTIC:
            if not in_valid_domain(m, sig) or m.startswith(RCFAM):
                continue
            ret, c = ref.press(m, sig, cap=int(ref.bound(m, len(sig))) + 16 * len(sig) + 64)
            assert ret == 0, (name, m, ret)
            # quirk 3: shuffman_*_depress works in a 2n-byte buffer (press.c:4472) - only ask the
            # reference to decode what fits (the oracle's own round trip covers the rest)
            plain = m.replace("shuffman_", "")
            fits = (not m.startswith("shuffman")) or len(ent["streams"][plain]) // 2 - 2 <= 2 * len(sig)
            if fits:
                r2, back = ref.depress(m, c, len(sig))
                assert r2 == 0 and np.array_equal(back, sig), (name, m)
            ent["streams"][m] = c.hex()
        micro.append(ent)
    json.dump({"source": "streams produced by the reference compiled from /root/reference/press",
               "vectors": micro}, open(os.path.join(HERE, "micro_kats.json"), "w"), indent=1)

    # ---- 3. svb32 issue42 -----------------------------------------------------------
    src = open(os.path.join(REF, "press/streamvbyte/tests/unit.c")).read()
    body = src[src.index("uint8_t a[36494] = {") + len("uint8_t a[36494] = {"):]
    body = body[:body.index("};")]
    a = bytes(int(x) for x in re.findall(r"\d+", body))
    assert len(a) == 36494
    open(os.path.join(HERE, "svb32_issue42.bin"), "wb").write(a)

    # ---- 4. synthetic reads ----------------------------------------------------------
    synth_cases = []
    seed = 20261004
    n_full, first = synth.read_lengths(seed, 0, 6)
    lens = [int(x) for x in n_full[:4]] + [1, 2, 7, 8, 9, 63, 64, 65, 511, 512, 513, 2047, 2048, 2049,
                                         4095, 4096, 4097, 8191, 8192, 8193, 16383, 16384, 16385, 40000, 65536 + 3]
    for k, n in enumerate(lens):
        r = k
        f = int(first[k % len(first)])
        sig = synth.synth_read(seed, r, n, f)
        ent = {"seed": seed, "read": r, "n": int(n), "first": f, "sha256_32_signal": sha(sig.tobytes()),
               "methods": {}}
        for m in _libs.METHODS:
            if not in_valid_domain(m, sig) or (m.startswith(RCFAM) and n < 64):
                continue
            ret, c = ref.press(m, sig)
            assert ret == 0, (m, n)
            e = {"len": len(c)}
            if m in _libs.DETERMINISTIC:
                e["sha256_32"] = sha(c)
            ent["methods"][m] = e
        synth_cases.append(ent)
    json.dump({"source": "honours_amd.synth reads; streams by the reference", "cases": synth_cases},
              open(os.path.join(HERE, "synth_kats.json"), "w"), indent=1)
    print("golden vectors written to", HERE, file=sys.stdout)


if __name__ == "__main__":
    try:
        main()
    except BaseException:
        import traceback
        traceback.print_exc(file=sys.stdout)  # stderr is silenced (reference diagnostics)
        sys.exit(1)

"""GPU: the reference's UNMODIFIED harness (press/test.c test_X functions, TEST() macro,
slow5lib loader - compiled in the dev container by `make -C oracle harness`, linked only
against libpress_hip.so) runs on data/three-reads.blow5 and reproduces the press_bytes
column of the reference's own run (BASELINE.md section 2 / SURVEY.md 8(c))."""
import os
import shutil
import subprocess

import pytest

from honours_amd import press

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "press_test_hip")

# press_bytes of the reference's own ./test ../data/three-reads.blow5 (deterministic methods)
GOLDEN = {
    "svb_zd": 322344, "svb12_zd": 290111, "vbe21_zd": 257924, "vbbe21_zd": 257919,
    "vbsse21_zd": 257921, "shuffman_vbe21_zd": 175227, "shuffman_vbbe21_zd": 175222,
    "shuffman_vbsse21_zd": 175224, "hasgam_vbsse21_zdq": 257953,
}
# libzstd-version dependent (parity unpinned): reference with zstd 1.4.9
ZSTD = {"zstd_svb_zd": 176598, "zstd_svb12_zd": 176612, "zstd_hasgam_vbsse21_zdq": 176660}


@pytest.mark.skipif(not os.path.exists(EXE), reason="oracle/_ref/press_test_hip not built (needs /root/reference)")
def test_reference_harness_on_hip_library(tmp_path):
    shutil.copy(press.TABLE_PATH, tmp_path / "NA12878_zd.huffman")  # opened by relative name, test.c:3786
    blow5 = os.path.join(ROOT, "tests", "golden", "three-reads.blow5")
    p = subprocess.run([EXE, blow5], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    rows = [l.split("\t") for l in p.stdout.strip().splitlines()]
    assert rows[0][:3] == ["method", "pressbound_bytes", "press_bytes"]
    got = {r[0]: r for r in rows[1:]}
    assert len(got) == 14
    for m, want in GOLDEN.items():
        assert int(float(got[m][2])) == want, (m, got[m])
        assert int(float(got[m][4])) == 515728  # depress_bytes = raw bytes of the three reads
    for m, want in ZSTD.items():
        assert abs(int(float(got[m][2])) - want) <= 0.01 * want, (m, got[m])

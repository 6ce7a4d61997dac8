#!/usr/bin/env python3
"""Wave-level instruction counts per launch of one method's kernels from one rocprofv3 counter pass of bench.py
(SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES) -> gpurun_out/<round>_<method>_valu.json (copy into profiles/:
bench.py's roofline.valu reads it).

    python3 tools/valu.py METHOD [ROUND_TAG]        (on the GPU box)

The VALU roofline of bench.py: a wave64 VALU instruction occupies its SIMD for 4 cycles when one wave issues alone;
CDNA4's SIMDs are 32 lanes wide, so with several waves per SIMD the pipe takes one instruction per 2 cycles:
peak = 256 CUs x 4 SIMDs x 2.4 GHz / 2 = 1229 G wave-instructions/s (MI355X_MICROARCH.md, 'vector-instruction ISSUE
cost')."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.traffic import PRESS  # noqa: E402

TIMED = {  # the kernels bench.py brackets with HIP events (press_hip_kernel_timing)
    "shuffman_vbe21_zd": (("k_huff_encode_chunked",), ("k_huf_sync", "k_huf_list", "k_huf_fix", "k_huf_serial", "k_huf_chain", "k_huf_emit")),
}


def main():
    method = sys.argv[1]
    tag = sys.argv[2] if len(sys.argv) > 2 else "r03"
    d = os.path.join(ROOT, "gpurun_out", "valu_%s" % method)
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_BUSY_CYCLES",
                    "--output-format", "csv", "-d", d, "--", "python3", os.path.join(ROOT, "bench.py"), "--method", method,
                    "--steps", "2", "--warmup", "1", "--no-cpu", "--no-sub"], check=True, env=env, stdout=subprocess.PIPE,
                   stderr=subprocess.PIPE, cwd=ROOT, timeout=600)
    f = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "ph::" not in k:
            continue
        k = k.replace("(anonymous namespace)::", "").split("(")[0].replace("void ph::", "").replace("ph::", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    per, tot = {}, collections.defaultdict(float)
    timed_p, timed_d = TIMED.get(method, ((), ()))
    for k, v in sorted(agg.items()):
        calls = max(1, round(len(disp[k]) / 3))  # the pass ran 3 calls
        n = len(disp[k])
        side = "press" if k.startswith(PRESS) else "depress"
        per[k] = {c: int(x / n * calls) for c, x in v.items()}
        per[k].update(launches_per_call=calls, side=side)
        tot[side] += v.get("SQ_INSTS_VALU", 0.0) / n * calls
        if k.startswith(timed_p):
            tot["press_timed"] += v.get("SQ_INSTS_VALU", 0.0) / n * calls
        if k.startswith(timed_d):
            tot["depress_timed"] += v.get("SQ_INSTS_VALU", 0.0) / n * calls
    out = {
        "source": "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES -- python3 bench.py "
                  "--method %s --steps 2 --warmup 1 --no-cpu --no-sub, MI355X (tools/valu.py); per call = per-launch "
                  "average x launches per call" % method,
        "workload": {"method": method, "reads_per_gpu": 8192, "seed": 20261004, "fixed_len": None},
        "kernels": per,
        "valu_insts_per_call": {k: int(v) for k, v in tot.items()},
    }
    p = os.path.join(ROOT, "gpurun_out", "%s_%s_valu.json" % (tag, method))
    json.dump(out, open(p, "w"), indent=1)
    print(json.dumps({"method": method, "valu": out["valu_insts_per_call"]}))


if __name__ == "__main__":
    main()

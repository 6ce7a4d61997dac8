#!/usr/bin/env python3
"""HBM traffic per launch of one method's kernels from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE in
separate runs - they do not fit one pass, MI355X_MICROARCH.md 'rocprofv3 PMC slots') of bench.py, corrected
as that guide's HBM section prescribes, -> profiles/r02_<method>_traffic.json (bench.py's roofline.traffic).

    python3 tools/traffic.py METHOD        (on the GPU box: gpurun -- python3 tools/traffic.py METHOD)

Units: the raw counters are KiB.  gfx950 corrections: FETCH_SIZE counts 1/2 of the bytes of 16-byte-per-lane
loads (x2; calibrated on tools/ubench_unaligned.hip in round 1: copy16 x2.000, unaligned 8-byte loads x1.890);
WRITE_SIZE is exact for 16-byte stores (8-byte unaligned stores: /1.028).  Infinity-Cache hits are counted as
traffic by these counters, so a second read of a buffer shows up even when the MALL serves it.
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# which kernels load their bulk with unaligned 8-byte loads (x1.89); every other kernel: 16-byte loads (x2)
LOAD8 = ("k_svb_decode_chunked", "k_low_decode_chunked")
STORE8 = ("k_svb_encode_chunked", "k_low_encode_chunked")
PRESS = ("k_chunk_prep<", "k_svb_encode", "k_ex_scan", "k_ex_prefix", "k_ex_list", "k_ex_section", "k_ex_fill",
         "k_low_encode", "k_huff_encode", "k_ex_redo", "k_zs_layout", "k_zs_blocks", "k_zs_blockmap", "k_zs_hist",
         "k_zs_keycount", "k_zs_table", "k_zs_keylist", "k_zs_bits", "k_zs_plan", "k_zs_encode", "k_zs_rawframes",
         "k_rcs_encode", "k_rcc_encode", "k_rcm_encode")


def run_pass(method, counter, tag):
    d = os.path.join(ROOT, "gpurun_out", "traffic_%s_%s" % (method, tag))
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "--",
                    "python3", os.path.join(ROOT, "bench.py"), "--method", method, "--steps", "2", "--warmup", "1",
                    "--no-cpu", "--no-sub"], check=True, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                   cwd=ROOT, timeout=600)  # (an abort under the profiler must end the call, not hang the box)
    f = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))[0]
    agg, disp = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "ph::" not in k:
            continue
        k = k.replace("(anonymous namespace)::", "").split("(")[0].replace("void ph::", "").replace("ph::", "")
        agg[k] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    return {k: v / len(disp[k]) for k, v in agg.items()}, {k: len(v) for k, v in disp.items()}


def main():
    method = sys.argv[1]
    fetch, nl = run_pass(method, "FETCH_SIZE", "fetch")
    write, _ = run_pass(method, "WRITE_SIZE", "write")
    bench = json.loads(subprocess.run(["python3", os.path.join(ROOT, "bench.py"), "--method", method, "--steps", "2",
                                       "--warmup", "1", "--no-cpu", "--no-sub"], check=True, stdout=subprocess.PIPE,
                                      cwd=ROOT).stdout.decode().strip().splitlines()[-1])
    per = {}
    tot = {"press": 0.0, "depress": 0.0}
    for k in sorted(set(fetch) | set(write)):
        ff = 1.89 if k.startswith(LOAD8) else 2.0
        wf = 1.0 / 1.028 if k.startswith(STORE8) else 1.0
        b = (fetch.get(k, 0.0) * ff + write.get(k, 0.0) * wf) * 1024.0
        # launches per call of this kernel (e.g. two repair rounds): the pass ran 3 calls
        calls = max(1, round(nl.get(k, 3) / 3))
        side = "press" if k.startswith(PRESS) else "depress"
        per[k] = {"FETCH_SIZE_KiB": round(fetch.get(k, 0.0), 1), "WRITE_SIZE_KiB": round(write.get(k, 0.0), 1),
                  "fetch_factor": ff, "launches_per_call": calls, "bytes_per_call": int(b * calls), "side": side}
        tot[side] += b * calls
    alg = bench["roofline"]["algorithmic_bytes_per_call"]
    out = {
        "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) -- python3 bench.py "
                  "--method %s --steps 2 --warmup 1 --no-cpu --no-sub, MI355X (tools/traffic.py); per-launch "
                  "averages, summed over the kernels of a call" % method,
        "workload": {"method": method, "reads_per_gpu": bench["config"]["reads_per_gpu"], "seed": 20261004,
                     "fixed_len": None},
        "kernels": per,
        "traffic_bytes_per_launch": {"press": int(tot["press"]), "depress": int(tot["depress"])},
        "algorithmic_bytes_per_call": alg,
        "traffic_over_algorithmic": {"press": round(tot["press"] / alg, 3), "depress": round(tot["depress"] / alg, 3)},
    }
    p = os.path.join(ROOT, "gpurun_out", "%s_%s_traffic.json" % (os.environ.get("ROUND_TAG", "r03"), method))
    json.dump(out, open(p, "w"), indent=1)
    print(json.dumps({"method": method, "traffic": out["traffic_bytes_per_launch"], "alg": alg,
                      "ratio": out["traffic_over_algorithmic"]}))


if __name__ == "__main__":
    main()

#!/bin/bash
# usage: tools/r3_zstraf.sh name... : k_zs_hdecode's FETCH_SIZE / WRITE_SIZE with tools/bin/libpress_<name>.so
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for v in "$@"; do
  for c in FETCH_SIZE WRITE_SIZE; do
    PRESS_HIP_LIB=tools/bin/libpress_$v.so timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/zt_${v}_$c -- python3 bench.py --method zstd_svb_zd --steps 2 --warmup 1 --no-cpu --no-sub > /dev/null 2> gpurun_out/zt_${v}_$c.err
    python3 - "$v" "$c" <<'PY'
import csv,glob,sys,collections
v,c=sys.argv[1:3]
f=glob.glob("gpurun_out/zt_%s_%s/*/*counter_collection.csv"%(v,c))[0]
tot=collections.defaultdict(float); n=collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"]
    if "k_zs_hdecode" in k:
        tot[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
for k in tot: print(v, k, "KiB per launch %.0f" % (tot[k]/len(n[k])))
PY
  done
done

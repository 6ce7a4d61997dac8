#!/bin/bash
# usage: tools/pmc_any.sh NAME "COUNTERS" -- program args...   (per-kernel per-launch averages, all kernels)
cd "$(dirname "$0")/.."
name=$1; ctrs=$2; shift 3
export TMPDIR=/tmp
timeout -k 10 ${PROF_TIMEOUT:-600} rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d gpurun_out/pmc_$name -- "$@" > gpurun_out/pmc_$name.out 2> gpurun_out/pmc_$name.err
python3 - "$name" <<'PY'
import csv,glob,collections,sys
f=glob.glob("gpurun_out/pmc_%s/*/*counter_collection.csv"%sys.argv[1])[0]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); disp=collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].split("(")[0].replace("void ","")
    if k.startswith("at::") or "rocprim" in k or "elementwise" in k or "Fill" in k: continue
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
for k,v in agg.items():
    n=len(disp[k])
    print("%-50s n=%d " % (k[:50], n) + " ".join("%s=%.1f" % (c, x/n) for c,x in sorted(v.items())))
PY

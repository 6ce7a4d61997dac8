#!/bin/bash
# usage: tools/pmc.sh NAME "COUNTER ..." [bench args]   -> per-kernel per-launch averages of ph:: kernels
cd "$(dirname "$0")/.."
name=$1; ctrs=$2; shift 2
export TMPDIR=/tmp
timeout -k 10 ${PROF_TIMEOUT:-600} rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d gpurun_out/pmc_$name -- python3 bench.py --steps 2 --warmup 1 --no-cpu "$@" > gpurun_out/pmc_$name.json 2> gpurun_out/pmc_$name.err
python3 - "$name" <<'PY'
import csv,glob,collections,sys
f=glob.glob("gpurun_out/pmc_%s/*/*counter_collection.csv"%sys.argv[1])[0]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); disp=collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"]
    if "ph::" not in k: continue
    k=k.replace("(anonymous namespace)::","").split("(")[0].replace("void ph::","").replace("ph::","")
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
for k,v in agg.items():
    n=len(disp[k])
    print(k, "launches", n)
    for c,x in sorted(v.items()): print("    %-32s %16.0f" % (c, x/n))
PY

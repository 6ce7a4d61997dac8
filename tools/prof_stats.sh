#!/bin/bash
# usage: tools/prof_stats.sh NAME -- program args...  (rocprofv3 kernel stats -> gpurun_out/NAME_stats.txt)
cd "$(dirname "$0")/.."
name=$1; shift 2
export TMPDIR=/tmp
timeout -k 10 ${PROF_TIMEOUT:-600} rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$name -- "$@" > gpurun_out/prof_$name.out 2> gpurun_out/prof_$name.err
f=$(ls gpurun_out/prof_$name/*/*kernel_stats.csv | head -1)
cp "$f" gpurun_out/${name}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Name"]
    if k.startswith("void at::") or "rocprim" in k or "elementwise" in k.lower() or "Fill" in k: continue
    print("%-70s calls %4s avg_us %10.1f total_ms %8.2f" % (k[:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY

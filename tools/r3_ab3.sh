#!/bin/bash
# usage: tools/r3_ab3.sh METHOD name... : bench of METHOD with tools/bin/libpress_<name>.so, one line each
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
m=$1; shift
for v in "$@"; do
  PRESS_HIP_LIB=tools/bin/libpress_$v.so timeout -k 10 120 python bench.py --method $m --no-sub --no-cpu --steps 20 --warmup 3 > gpurun_out/ab3_$v.json 2> gpurun_out/ab3_$v.err || { echo "$v failed"; tail -3 gpurun_out/ab3_$v.err; exit 1; }
  python3 -c "
import json;d=json.load(open('gpurun_out/ab3_$v.json'));print('$m','$v','value',d['value'],'press',d['roofline_other']['avg_launch_ms'],'/',d['roofline_other']['whole_call_ms'],'depress',d['roofline']['avg_launch_ms'],'/',d['roofline']['whole_call_ms'])"
done

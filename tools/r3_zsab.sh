#!/bin/bash
# usage: tools/r3_zsab.sh name... : config 3 bench with tools/bin/libpress_<name>.so, one line each (+ kernel times)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for v in "$@"; do
  PRESS_HIP_LIB=tools/bin/libpress_$v.so timeout -k 10 200 python bench.py --method zstd_svb_zd --no-sub --no-cpu --steps 20 --warmup 3 > gpurun_out/zsab_$v.json 2> gpurun_out/zsab_$v.err || { echo "$v failed"; tail -3 gpurun_out/zsab_$v.err; exit 1; }
  python3 -c "
import json;d=json.load(open('gpurun_out/zsab_$v.json'));print('$v','value',d['value'],'press',d['roofline_other']['whole_call_ms'],'depress',d['roofline']['whole_call_ms'],'hdecode',d['roofline']['avg_launch_ms'])"
done

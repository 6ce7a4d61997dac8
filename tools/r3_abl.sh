#!/bin/bash
# timing experiments with ablated kernels (tools/bin/libpress_<v>.so; wrong results, --no-check): kernel stats per variant
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for v in "$@"; do
  lib=$PWD/tools/bin/libpress_$v.so; [ "$v" = base ] && lib=$PWD/honours_amd/libpress_hip.so
  PRESS_HIP_LIB=$lib PROF_TIMEOUT=150 bash tools/prof_stats.sh abl_$v -- python3 bench.py --no-sub --no-cpu --no-check --steps 5 --warmup 2 > gpurun_out/abl_${v}_stats.txt 2>&1 || { echo "variant $v failed"; tail -3 gpurun_out/prof_abl_$v.err; continue; }
  echo "== $v  $(grep -E 'k_huf_emit|k_huf_sync' gpurun_out/abl_${v}_stats.txt | awk '{print $2, $(NF-2)}' | tr '\n' ' ')"
done

#!/bin/bash
# round 3, first GPU call: default bench line, run-up variants of k_huf_sync (kernel stats), then the GPU suite
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 420 python bench.py --steps 20 --warmup 3 > gpurun_out/r3_bench1.json 2> gpurun_out/r3_bench1.err || { echo "bench failed"; tail -5 gpurun_out/r3_bench1.err; exit 1; }
echo "bench done"
for v in base ru3 ru2 ru2f6; do
  PRESS_HIP_LIB=$PWD/tools/bin/libpress_$v.so PROF_TIMEOUT=150 bash tools/prof_stats.sh r3_$v -- python3 bench.py --no-sub --no-cpu --steps 5 --warmup 2 > gpurun_out/r3_${v}_stats.txt 2>&1 || { echo "variant $v failed"; exit 1; }
  echo "== $v"; grep -E "k_huf|k_huff" gpurun_out/r3_${v}_stats.txt
done
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r3_tests1.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r3_tests1.log

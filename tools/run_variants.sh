#!/bin/bash
# run bench.py against each diagnostic library; print press launch time
cd "$(dirname "$0")/.."
for lib in "$@"; do
  if [ "$lib" = "base" ]; then unset PRESS_HIP_LIB; else export PRESS_HIP_LIB=$PWD/tools/bin/libpress_$lib.so; fi
  out=$(timeout -k 10 120 python bench.py --steps 5 --warmup 2 --no-cpu --no-check 2>/dev/null)
  echo "$lib $(echo "$out" | python -c 'import sys,json; d=json.load(sys.stdin); print("press_ms", d["roofline"]["avg_launch_ms"], "GB/s", d["roofline"]["achieved"], "depress_ms", d["roofline_depress"]["avg_launch_ms"])')"
done

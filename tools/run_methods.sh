#!/bin/bash
cd "$(dirname "$0")/.."
for m in "$@"; do
  out=$(timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu --method $m --reads 4096 2>/dev/null)
  echo "$m $(echo "$out" | python -c 'import sys,json; d=json.load(sys.stdin); print("value", d["value"], "ratio", d["ratio"], "press_ms", d["roofline"]["avg_launch_ms"], "GB/s", d["roofline"]["achieved"], "depress_ms", d["roofline_depress"]["avg_launch_ms"], "GB/s", d["roofline_depress"]["achieved"])')"
done

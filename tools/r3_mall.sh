#!/bin/bash
# does the Huffman encoder run faster when pass A has just left its input in the Infinity Cache?  per-kernel times at
# batch sizes below and above the 256-MB cache
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for r in 256 512 1024 2048 8192; do
  PROF_TIMEOUT=150 bash tools/prof_stats.sh mall_$r -- python3 bench.py --no-sub --no-cpu --steps 10 --warmup 2 --reads $r > gpurun_out/mall_${r}_stats.txt 2>&1
  raw=$(python3 -c "import json;print(json.load(open('gpurun_out/prof_mall_$r.out'))['config']['samples_per_gpu']*2/1e6)")
  echo "== $r reads, $raw MB raw: $(grep -E 'k_huff_encode|k_ex_scan' gpurun_out/mall_${r}_stats.txt | awk '{print $1, $(NF-2)}' | tr '\n' ' ')"
done

#!/bin/bash
# Huffman kernels: natural read-length mix vs. fixed-length reads of the same total size
cd "$(dirname "$0")/.."
p() { python -c 'import sys,json; d=json.load(sys.stdin); a=d["roofline"]; b=d["roofline_other"]; print("value", d["value"], "|", a["kernel"], a["avg_launch_ms"], "call", a["whole_call_ms"], "|", b["kernel"], b["avg_launch_ms"], "call", b["whole_call_ms"])'; }
echo "natural 8192:"; timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu --method shuffman_vbe21_zd --reads 8192 2>/dev/null | p
echo "fixed 113500 x 8192:"; timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu --method shuffman_vbe21_zd --reads 8192 --fixed-len 113500 2>/dev/null | p
echo "fixed 14000 x 65536:"; timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu --method shuffman_vbe21_zd --reads 65536 --fixed-len 14000 2>/dev/null | p

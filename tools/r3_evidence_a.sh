#!/bin/bash
# round-3 evidence, part A: counter passes (HBM traffic, VALU instruction counts, SQ counters) of the final kernels
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp ROUND_TAG=r03
for m in shuffman_vbe21_zd svb12_zd vbe21_zd zstd_svb_zd; do
  timeout -k 10 400 python3 tools/traffic.py $m || echo "traffic $m failed"
done
timeout -k 10 300 python3 tools/valu.py shuffman_vbe21_zd r03 || echo "valu failed"
bash tools/huf_prof.sh r03_shuf "k_huf_sync|k_huf_emit|k_huff_encode|k_ex_scan" > gpurun_out/r03_shuffman_vbe21_zd_pmc.txt 2>&1
tail -3 gpurun_out/r03_shuffman_vbe21_zd_pmc.txt

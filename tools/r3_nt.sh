#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout -k 5 90 python __graft_entry__.py smoke > gpurun_out/safe_smoke.log 2>&1 || { echo smoke failed; tail -3 gpurun_out/safe_smoke.log; exit 1; }
for v in nont base; do
  lib=$PWD/tools/bin/libpress_$v.so; [ "$v" = base ] && lib=$PWD/honours_amd/libpress_hip.so
  for m in shuffman_vbe21_zd svb12_zd; do
    rm -rf gpurun_out/prof_nt_${v}_$m
    PRESS_HIP_LIB=$lib PROF_TIMEOUT=150 bash tools/prof_stats.sh nt_${v}_$m -- python3 bench.py --no-sub --no-cpu --steps 10 --warmup 2 --method $m > gpurun_out/nt_${v}_${m}_stats.txt 2>&1
    echo "== $v $m: $(grep -E 'k_ex_scan|k_huff_encode|k_svb_encode|k_svb_decode|k_huf_sync|k_huf_emit' gpurun_out/nt_${v}_${m}_stats.txt | awk '{print substr($1,1,28), $(NF-2)}' | tr '\n' ' ') value $(python3 -c "import json;print(json.load(open('gpurun_out/prof_nt_${v}_$m.out'))['value'])")"
  done
done

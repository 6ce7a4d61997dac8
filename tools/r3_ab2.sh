#!/bin/bash
# kernel stats of svb12_zd (config 2) and the headline with the in-tree library, behind the parity subset
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout -k 5 90 python __graft_entry__.py smoke > gpurun_out/safe_smoke.log 2>&1 || { echo smoke failed; tail -3 gpurun_out/safe_smoke.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/ab2_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -2 gpurun_out/ab2_tests.log; [ $rc -eq 0 ] || exit 1
for m in svb12_zd shuffman_vbe21_zd; do
  PROF_TIMEOUT=150 bash tools/prof_stats.sh ab2_$m -- python3 bench.py --no-sub --no-cpu --steps 10 --warmup 2 --method $m > gpurun_out/ab2_${m}_stats.txt 2>&1
  grep -E "k_svb_decode|k_svb_encode|k_svb_key|k_huff_encode|k_ex_scan|k_huf_emit|k_huf_sync" gpurun_out/ab2_${m}_stats.txt
  python3 -c "
import json;d=json.load(open('gpurun_out/prof_ab2_$m.out'));print('$m value',d['value'],'press',d['roofline_other']['whole_call_ms'] if d['roofline']['kernel'].find('decode')>=0 or d['roofline']['kernel'].find('huf_sync')>=0 else d['roofline']['whole_call_ms'])"
done

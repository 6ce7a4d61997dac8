#!/bin/bash
# usage: tools/r3_zs.sh TAG : config 3 (zstd_svb_zd) - smoke, GPU parity subset, kernel stats, bench line
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
tag=$1
timeout -k 5 90 python __graft_entry__.py smoke > gpurun_out/safe_smoke.log 2>&1; rc=$?
echo "smoke rc=$rc"; tail -1 gpurun_out/safe_smoke.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "zstd or three_reads" > gpurun_out/${tag}_tests.log 2>&1; rc=$?
echo "tests rc=$rc"; tail -4 gpurun_out/${tag}_tests.log
[ $rc -eq 0 ] || exit 1
PROF_TIMEOUT=150 bash tools/prof_stats.sh ${tag} -- python3 bench.py --method zstd_svb_zd --no-sub --no-cpu --steps 5 --warmup 2 > gpurun_out/${tag}_stats.txt 2>&1 || { echo "profile failed"; exit 1; }
grep -E "k_zs|k_svb" gpurun_out/${tag}_stats.txt
timeout -k 10 200 python bench.py --method zstd_svb_zd --no-sub --no-cpu --steps 20 --warmup 3 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err && python3 -c "
import json;d=json.load(open('gpurun_out/${tag}_bench.json'));print('value',d['value'],'press',d['roofline_other']['whole_call_ms'],'depress',d['roofline']['whole_call_ms'],'ratio',d.get('ratio'))"

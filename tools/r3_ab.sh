#!/bin/bash
# usage: tools/r3_ab.sh TAG [pytest -k expr] : GPU parity subset, then kernel stats of the headline with the in-tree library
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
tag=$1; kexpr=${2:-huff or shuff or bench_batch or three_reads or config5}
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "$kexpr" > gpurun_out/${tag}_tests.log 2>&1; rc=$?
echo "tests rc=$rc"; tail -4 gpurun_out/${tag}_tests.log
[ $rc -eq 0 ] || exit 1
PROF_TIMEOUT=150 bash tools/prof_stats.sh ${tag} -- python3 bench.py --no-sub --no-cpu --steps 5 --warmup 2 > gpurun_out/${tag}_stats.txt 2>&1 || { echo "profile failed"; exit 1; }
grep -E "k_huf|k_huff|k_ex_scan" gpurun_out/${tag}_stats.txt
timeout -k 10 200 python bench.py --no-sub --no-cpu --steps 20 --warmup 3 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err && python3 -c "
import json;d=json.load(open('gpurun_out/${tag}_bench.json'));print('value',d['value'],'press',d['roofline_other']['whole_call_ms'],'depress',d['roofline']['whole_call_ms'],'kern',d['roofline']['avg_launch_ms'])"

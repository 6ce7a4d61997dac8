#!/bin/bash
# round-3 evidence, part E: the svb methods after k_svb_decode_chunked changed (surplus workgroups draw no ticket), and the
# default bench line
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
for m in svb12_zd svb_zd slow5_svb_zd; do
  timeout -k 10 300 python bench.py --method $m > gpurun_out/r03_final_bench_$m.json 2> gpurun_out/r03_final_bench_$m.err || { echo "$m bench failed"; exit 1; }
  tools/prof_stats.sh r03_final_$m -- python3 bench.py --steps 5 --warmup 2 --no-cpu --no-sub --method $m > gpurun_out/r03_final_${m}_stats.txt || { echo "$m profile failed"; exit 1; }
  echo "$m done"
done
bash tools/r3_evidence_d.sh

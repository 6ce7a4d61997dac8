#!/bin/bash
# end-of-round bench lines + rocprofv3 kernel stats for every method -> gpurun_out/final_*
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for m in ${METHODS:-svb12_zd svb_zd slow5_svb_zd vbe21_zd hasgam_vbsse21_zdq shuffman_vbe21_zd zstd_svb_zd}; do # rc_vbe21_zd: bench line only (profiles/README.md)
  extra="--method $m"
  timeout -k 10 300 python bench.py $extra > gpurun_out/final_bench_$m.json 2> gpurun_out/final_bench_$m.err || exit 1
  tools/prof_stats.sh final_$m -- python3 bench.py --steps 5 --warmup 2 --no-cpu $extra > gpurun_out/final_${m}_stats.txt || exit 1
  echo "$m done"
done

#!/bin/bash
# end-of-round evidence (run on the GPU box: gpurun -- bash tools/final_profiles.sh):
#   gpurun_out/${TAG:-r02_final}_bench.json                  the default bench line (headline + sub-records)
#   gpurun_out/${TAG:-r02_final}_bench_<method>.json         one line per method (with cpu_baseline)
#   gpurun_out/${TAG:-r02_final}_<method>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the same command
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout -k 10 400 python bench.py > gpurun_out/${TAG:-r02_final}_bench.json 2> gpurun_out/${TAG:-r02_final}_bench.err || exit 1
echo "default done"
for m in ${METHODS:-svb12_zd svb_zd slow5_svb_zd vbe21_zd hasgam_vbsse21_zdq shuffman_vbe21_zd zstd_svb_zd zstd_svb12_zd zstd_hasgam_vbsse21_zdq}; do
  extra="--method $m"
  timeout -k 10 300 python bench.py $extra > gpurun_out/${TAG:-r02_final}_bench_$m.json 2> gpurun_out/${TAG:-r02_final}_bench_$m.err || exit 1
  tools/prof_stats.sh ${TAG:-r02_final}_$m -- python3 bench.py --steps 5 --warmup 2 --no-cpu --no-sub $extra > gpurun_out/${TAG:-r02_final}_${m}_stats.txt || exit 1
  echo "$m done"
done
for m in rc_vbe21_zd rcc_vbe21_zd rccm_vbbe21_zd; do # serial per read: seconds per step - bench line only, two steps
  timeout -k 10 500 python bench.py --method $m --steps 2 --warmup 1 > gpurun_out/${TAG:-r02_final}_bench_$m.json 2> gpurun_out/${TAG:-r02_final}_bench_$m.err || exit 1
  echo "$m done"
done

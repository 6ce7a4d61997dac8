#!/bin/bash
# round-3 evidence, part C (config 3 after its kernels changed): traffic counters, bench lines + rocprofv3 kernel stats of the
# three zstd compositions, the reader on libzstd's own frames, the stamps of k_zs_hdecode / k_zs_walk / k_zs_table
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp ROUND_TAG=r03
timeout -k 10 400 python3 tools/traffic.py zstd_svb_zd || echo "traffic failed"
for m in zstd_svb_zd zstd_svb12_zd zstd_hasgam_vbsse21_zdq; do
  timeout -k 10 300 python bench.py --method $m > gpurun_out/r03_final_bench_$m.json 2> gpurun_out/r03_final_bench_$m.err || { echo "$m bench failed"; exit 1; }
  tools/prof_stats.sh r03_final_$m -- python3 bench.py --steps 5 --warmup 2 --no-cpu --no-sub --method $m > gpurun_out/r03_final_${m}_stats.txt || { echo "$m profile failed"; exit 1; }
  echo "$m done"
done
PROF_TIMEOUT=200 bash tools/prof_stats.sh r03_libzstd_frames -- python3 tools/lzframes.py 2048 5 > gpurun_out/r03_libzstd_frames_stats.txt 2>&1; echo "lzframes profile rc=$?"
grep -E "k_zs_" gpurun_out/r03_libzstd_frames_stats.txt | head -8; tail -2 gpurun_out/prof_r03_libzstd_frames.out
[ -f tools/bin/libpress_hstamp.so ] && timeout -k 10 200 python tools/zsstamps.py > gpurun_out/r03_zstd_stamps.txt 2>&1; tail -30 gpurun_out/r03_zstd_stamps.txt | grep -v "^{"

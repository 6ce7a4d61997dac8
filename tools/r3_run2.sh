#!/bin/bash
# round 3: the profile VERDICT item 3 asks for (the reference's own zstd frames through the device reader, under rocprofv3),
# and the order-0 range coder on a batch that fills the chip (item 5)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
PROF_TIMEOUT=200 bash tools/prof_stats.sh r03_libzstd_frames -- python3 tools/lzframes.py 2048 5 > gpurun_out/r03_libzstd_frames_stats.txt 2>&1; echo "lzframes profile rc=$?"; grep -E "k_zs_" gpurun_out/r03_libzstd_frames_stats.txt | head -12; tail -2 gpurun_out/prof_r03_libzstd_frames.out
timeout -k 10 600 python3 bench.py --method rc_vbe21_zd --reads 65536 --steps 2 --warmup 1 --no-sub > gpurun_out/r03_bench_rc_65536.json 2> gpurun_out/r03_bench_rc_65536.err; echo "rc bench rc=$?"; python3 -c "
import json;d=json.load(open('gpurun_out/r03_bench_rc_65536.json'));print('value',d['value'],'press',d['press_MBps'],'depress',d['depress_MBps'],'ratio',d['ratio'],'cpu',d.get('cpu_baseline',{}).get('all_cores'))"

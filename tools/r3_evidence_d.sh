#!/bin/bash
# round-3 evidence, part D: the default bench line (the driver's command) with the final kernels
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 500 python bench.py > gpurun_out/r03_final_bench.json 2> gpurun_out/r03_final_bench.err || { echo "default bench failed"; tail -5 gpurun_out/r03_final_bench.err; exit 1; }
python3 -c "
import json;d=json.load(open('gpurun_out/r03_final_bench.json'));print('value',d['value'],'frac',d['roofline']['frac'],'configs',{k:(v.get('value'),v.get('ratio')) for k,v in d.get('configs',{}).items()})"

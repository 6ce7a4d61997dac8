#!/bin/bash
# usage: tools/huf_prof.sh TAG [kernel-regex]  - kernel stats + two SQ counter passes of the headline method (gpurun box)
cd "$(dirname "$0")/.."
tag=$1; pat=${2:-k_huf}
bash tools/prof_stats.sh ${tag} -- python3 bench.py --no-sub --no-cpu --steps 3 --warmup 1 > gpurun_out/${tag}_stats.txt 2>&1
bash tools/pmc.sh ${tag}_a "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" --no-sub > gpurun_out/${tag}_pmc_a.txt 2>&1
bash tools/pmc.sh ${tag}_b "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" --no-sub > gpurun_out/${tag}_pmc_b.txt 2>&1
grep -E "k_huf|k_low_decode|k_huff" gpurun_out/${tag}_stats.txt
grep -A8 -E "$pat" gpurun_out/${tag}_pmc_a.txt | head -40
grep -A9 -E "$pat" gpurun_out/${tag}_pmc_b.txt | head -40

#!/bin/bash
# usage: tools/pmc2.sh TAG METHOD KERNEL-REGEX  - two SQ counter passes + one TCP/TCC pass for a method's kernels
cd "$(dirname "$0")/.."
tag=$1; m=$2; pat=$3
bash tools/pmc.sh ${tag}_a "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" --no-sub --method $m > gpurun_out/${tag}_pmc_a.txt 2>&1
bash tools/pmc.sh ${tag}_b "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" --no-sub --method $m > gpurun_out/${tag}_pmc_b.txt 2>&1
grep -A8 -E "$pat" gpurun_out/${tag}_pmc_a.txt | head -60
grep -A9 -E "$pat" gpurun_out/${tag}_pmc_b.txt | head -60

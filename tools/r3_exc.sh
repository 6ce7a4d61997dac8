#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for m in vbe21_zd hasgam_vbsse21_zdq; do for r in 0.0 0.3; do
  PROF_TIMEOUT=150 bash tools/prof_stats.sh exc_${m}_$r -- python3 tools/exc_heavy.py $r $m > gpurun_out/exc_${m}_${r}_stats.txt 2>&1
  echo "== $m rate $r: $(tail -1 gpurun_out/prof_exc_${m}_$r.out)"; grep -v "copyBuffer\|fillBuffer" gpurun_out/exc_${m}_${r}_stats.txt | sort -k8 -n -r -t' ' | awk '{ if ($(NF-2)+0 > 60) print }' | head -12
done; done

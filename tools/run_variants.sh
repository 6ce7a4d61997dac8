#!/bin/bash
# run bench.py against each diagnostic library; print main-kernel and whole-call times
cd "$(dirname "$0")/.."
for lib in "$@"; do
  if [ "$lib" = "base" ]; then unset PRESS_HIP_LIB; else export PRESS_HIP_LIB=$PWD/tools/bin/libpress_$lib.so; fi
  out=$(timeout -k 10 120 python bench.py --steps 5 --warmup 2 --no-cpu --no-check 2>/dev/null)
  echo "$lib $(echo "$out" | python -c 'import sys,json; d=json.load(sys.stdin); a=d["roofline"]; b=d["roofline_other"]; print(a["kernel"], a["avg_launch_ms"], "call", a["whole_call_ms"], "|", b["kernel"], b["avg_launch_ms"], "call", b["whole_call_ms"])')"
done

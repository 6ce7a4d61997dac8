#!/bin/bash
cd "$(dirname "$0")/.."
R=${READS:-4096}
for m in "$@"; do
  out=$(timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu --method $m --reads $R 2>/dev/null)
  echo "$m $(echo "$out" | python -c 'import sys,json; d=json.load(sys.stdin); a=d["roofline"]; b=d["roofline_other"]; print("value", d["value"], "ratio", d["ratio"], "press_MBps", d["press_MBps"], "depress_MBps", d["depress_MBps"], "|", a["kernel"], a["avg_launch_ms"], "call", a["whole_call_ms"], "|", b["kernel"], b["avg_launch_ms"], "call", b["whole_call_ms"])')"
done

#!/bin/bash
# smoke under a short timeout first (a hang costs a minute, not the call), then the A/B run
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 5 90 python __graft_entry__.py smoke > gpurun_out/safe_smoke.log 2>&1; rc=$?
echo "smoke rc=$rc"; tail -3 gpurun_out/safe_smoke.log
[ $rc -eq 0 ] || exit 1
bash tools/r3_ab.sh "$@"

#!/bin/bash
# the whole GPU suite (behind a smoke run under a short timeout), optionally followed by "$@"
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 5 90 python __graft_entry__.py smoke > gpurun_out/safe_smoke.log 2>&1; rc=$?
echo "smoke rc=$rc"; [ $rc -eq 0 ] || { tail -5 gpurun_out/safe_smoke.log; exit 1; }
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/full_tests.log 2>&1; rc=$?
echo "tests rc=$rc"; tail -4 gpurun_out/full_tests.log
[ $rc -eq 0 ] || exit 1
"$@"

#!/bin/bash
# round-3 evidence, part B: the default bench line and per-method lines + rocprofv3 kernel stats (tools/final_profiles.sh)
cd "$(dirname "$0")/.."
TAG=r03_final METHODS="${METHODS:-shuffman_vbe21_zd svb12_zd zstd_svb_zd vbe21_zd hasgam_vbsse21_zdq svb_zd slow5_svb_zd}" bash tools/final_profiles.sh

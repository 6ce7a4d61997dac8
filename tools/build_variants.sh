#!/bin/bash
# diagnostic builds of libpress_hip.so with experiment switches -> tools/bin/libpress_<name>.so
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/bin
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-cast-align -Wno-unused-function $flags \
     honours_amd/csrc/press_kernels.hip honours_amd/csrc/press_chunked.hip honours_amd/csrc/press_huffman.hip honours_amd/csrc/press_rc.hip honours_amd/csrc/press_zstd.hip honours_amd/csrc/press_abi.hip honours_amd/csrc/blow5_reader.cpp \
     -o tools/bin/libpress_$name.so -ldl &
done
wait
ls -la tools/bin/*.so

#!/bin/bash
# diagnostic builds of libpress_hip.so with experiment switches -> tools/bin/libpress_<name>.so
#   tools/build_variants.sh "name:-DFLAG=1 -DOTHER=2" ...      (run with PRESS_HIP_LIB=tools/bin/libpress_<name>.so)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/bin
srcs=$(python3 -c "import honours_amd.build as b, os; print(' '.join(os.path.join(b.CSRC, f) for f in b.SOURCES))")
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-cast-align -Wno-unused-function $flags \
     $srcs -o tools/bin/libpress_$name.so -ldl &
done
wait
ls -la tools/bin/*.so

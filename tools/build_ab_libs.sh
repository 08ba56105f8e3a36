#!/bin/bash
# Product-flag builds of libgumbi_hip.so with compile-time switches set, for tools/gpu_ab_libs.py:
#   tools/build_ab_libs.sh name1 "-DCT_STRIP_LDS=0 -DCT_RAGGED=0" name2 "-DCT_RAGGED=0" ...   -> gumbi_amd/lib/ab_<name>.so
# (the .so files are git-ignored and travel to the GPU box with the snapshot)
set -e
cd "$(dirname "$0")/../gumbi_amd/csrc"
while [ $# -ge 2 ]; do
  name=$1; defs=$2; shift 2
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-function $defs engine.hip -o ../lib/ab_${name}.so &
done
wait
ls -la ../lib/ab_*.so

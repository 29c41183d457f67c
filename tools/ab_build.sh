#!/bin/bash
# Build a variant of the library for an in-call A/B:  tools/ab_build.sh <name> <file.hip> "<extra hipcc flags>"
# -> raindrop_amd/_ab/lib_<name>.so (all objects of the normal build, with <file.hip> recompiled with the flags).
# Use on the GPU box as  RD_LIB_PATH=raindrop_amd/_ab/lib_<name>.so python bench.py ...
set -e
cd "$(dirname "$0")/.."
mkdir -p raindrop_amd/_ab
base=$(basename "$2" .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-value -Wno-unused-result -I include $3 \
  -c "$2" -o raindrop_amd/_ab/${base}_$1.o
objs=$(ls raindrop_amd/csrc/_build/*.o | grep -v "/${base}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o raindrop_amd/_ab/lib_$1.so $objs raindrop_amd/_ab/${base}_$1.o
echo raindrop_amd/_ab/lib_$1.so

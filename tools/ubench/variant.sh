#!/bin/bash
# Build an experimental variant of libzhilight_amd.so: one source recompiled with extra -D flags, the
# rest of the objects reused.  usage: variant.sh <name> <source.hip> -DFLAG...   -> zhilight_amd/build/variants/lib<name>.so
set -e
cd "$(dirname "$0")/../.."
name=$1; src=$2; shift 2
mkdir -p zhilight_amd/build/variants
base=$(basename "$src" .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -fno-fast-math -ffp-contract=off \
  "$@" -c "$src" -o zhilight_amd/build/variants/${name}_${base}.o
objs=$(ls zhilight_amd/build/*.o | grep -v "/${base}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o zhilight_amd/build/variants/lib${name}.so $objs zhilight_amd/build/variants/${name}_${base}.o
echo zhilight_amd/build/variants/lib${name}.so

#!/bin/bash
# build_variant.sh <name> <source.hip> <extra flags...>: scripts/variants/lib_<name>.so = the product library with one source
# rebuilt under extra flags (experiments: scripts/gpu_canny.sh etc. time every library under scripts/variants/)
set -e
cd "$(dirname "$0")/../image_amd/csrc"
name=$1; src=$2; shift 2
mkdir -p build/var ../../scripts/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -fvisibility=hidden -Wall -Wno-unused-function -DIMGFD_BUILD "$@" -c $src -o build/var/${name}_${src%.hip}.o
objs=$(ls build/*.o | grep -v "/emu_" | grep -v "build/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../../scripts/variants/lib_${name}.so $objs build/var/${name}_${src%.hip}.o -Wl,-Bsymbolic -Wl,-rpath,/opt/rocm/lib
echo built scripts/variants/lib_${name}.so

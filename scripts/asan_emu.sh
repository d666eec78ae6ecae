#!/bin/bash
# Sanitizers over the kernels: the library's sources compiled for the host-side HIP emulator with -fsanitize=address
# (global "device" buffers are malloc'ed: red zones around every one; LDS arrays are statics with red zones) or, SAN=undefined,
# with UBSan (shift widths, signed overflow, misaligned accesses ...), then the CPU test-suite against that build.
# Usage: [SAN=address|undefined] scripts/asan_emu.sh [pytest arguments]   (default: tests/)
set -e
SAN=${SAN:-address}
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/san_$SAN; mkdir -p $O
cd $R/image_amd/csrc
ls *.hip *.cpp | xargs -P 16 -I{} sh -c 'f={}; b=${f%.*}; x="-x c++ -fvisibility=hidden -I../../tests/hipemu -Wno-unknown-pragmas -DIMGFD_BUILD"; echo $f | grep -q "_host.cpp" && x=""; [ '$O'/$b.o -nt $f ] || g++ -O1 -g -std=c++17 -ffp-contract=off -fPIC -fsanitize='$SAN' -fno-sanitize-recover=all -fno-omit-frame-pointer $x -c $f -o '$O'/$b.o'
g++ -shared -fPIC -pthread -fsanitize=$SAN -Wl,-Bsymbolic -o $O/libimgfd_emu.so $O/*.o
cd $R
export IMGFD_EMU_LIB=$O/libimgfd_emu.so LD_PRELOAD=$(gcc -print-file-name=$([ $SAN = address ] && echo libasan.so || echo libubsan.so))
export ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
exec python -m pytest -m "not gpu" -q -p no:cacheprovider "${@:-tests}"

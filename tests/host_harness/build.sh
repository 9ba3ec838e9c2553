#!/bin/bash
# TEST INFRASTRUCTURE ONLY: build the kernel sources for the CPU through the HIP stand-in (see hip/hip_runtime.h).
set -e
CXX=/opt/rocm/lib/llvm/bin/clang++
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(dirname "$(dirname "$HERE")")"
SRCS=$(ls "$ROOT"/nerfool_amd/csrc/*.hip)
OBJS=""
mkdir -p "$HERE/build"
for s in $SRCS "$HERE/hip_emu.cpp"; do
  o="$HERE/build/$(basename "$s").o"
  if [ ! -f "$o" ] || [ "$s" -nt "$o" ] || [ "$HERE/hip/hip_runtime.h" -nt "$o" ] || [ -n "$(find "$ROOT/nerfool_amd/csrc" "$ROOT/include" -name '*.h' -newer "$o")" ]; then
    $CXX -O1 -std=c++17 -fPIC -pthread -ffp-contract=off -Wno-unknown-pragmas -Wno-unknown-attributes -Wno-pass-failed -Wno-psabi -x c++ -I "$HERE" -c "$s" -o "$o" &
  fi
  OBJS="$OBJS $o"
done
wait
$CXX -shared -pthread -o "$HERE/libnerfool_emu.so" $OBJS
echo "built $HERE/libnerfool_emu.so"

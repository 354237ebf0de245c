#!/bin/sh
# Builds oracle/_ref/libvsref.so: the REFERENCE's own scalar translation units, compiled from where they lie
# under /root/reference with plain g++ (not its CMake, no stand-in headers, no third-party code), plus
# oracle/ref_driver.cpp (our C entry points over them).  Test infrastructure only; outputs go to oracle/_ref/
# (git-ignored).  Flags follow the reference's own for these files: -std=gnu++20 (src/VecSim/CMakeLists.txt:15),
# -fPIC -fexceptions (CMakeLists.txt:35), no -m flags on L2.cpp / IP.cpp (they are the baseline-ISA "no
# optimisation" kernels), CMake's Release default -O3.
#
# Does nothing (exit 0) when /root/reference is absent: the GPU box uses the prebuilt file that travelled.
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
REF=${VECSIM_REFERENCE:-/root/reference}
SRC=$REF/src/VecSim
if [ ! -f "$SRC/spaces/L2/L2.cpp" ]; then
    echo "build_ref.sh: $REF not present, keeping whatever oracle/_ref holds"
    exit 0
fi
mkdir -p "$HERE/_ref"
OUT=$HERE/_ref/libvsref.so
if [ -f "$OUT" ] && [ "$OUT" -nt "$HERE/ref_driver.cpp" ] && [ "$OUT" -nt "$HERE/build_ref.sh" ]; then
    exit 0
fi
g++ -std=gnu++20 -O3 -DNDEBUG -fPIC -fexceptions -Wall -I"$REF/src" -shared -o "$OUT" \
    "$SRC/spaces/L2/L2.cpp" "$SRC/spaces/IP/IP.cpp" \
    "$SRC/memory/vecsim_malloc.cpp" "$SRC/memory/vecsim_base.cpp" \
    "$HERE/ref_driver.cpp"
echo "built $OUT"

#!/bin/bash
# Builds a DELIBERATELY BROKEN libltx2hip (every counted LDS wait of the attention kernel allows three more reads in flight than is safe: fragments
# are consumed before they land) into ltx-2-mlx_amd/lib/ab/broken_wait.so -- on the CPU box, before gpurun -- so that
#   LTX2HIP_LIB=$PWD/ltx-2-mlx_amd/lib/ab/broken_wait.so python -m pytest tests/test_kernels_gpu.py tests/test_parity.py -q -k "flash_attn or baseline_size_block"
# can show on the GPU that the tightened parity gates FAIL on it ("a deliberately broken build fails the suite").
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/ltx-2-mlx_amd/csrc
T=$(mktemp -d)
cp "$CSRC"/*.hip "$CSRC"/*.h "$CSRC"/*.inc "$T"/
mkdir -p "$T/../../include" 2>/dev/null || true
sed -i 's|#include "../../include/ltx2hip.h"|#include "'"$ROOT"'/include/ltx2hip.h"|' "$T"/*.hip "$T"/*.h
# the break: lds_wait<N> waits for lgkmcnt(N + 3)
sed -i 's|asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "n"(N) : "memory");|asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "n"(N + 3) : "memory");|' "$T/common.h"
grep -q 'N + 3' "$T/common.h"
make -C "$CSRC" -j8 > /dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-inline-asm -fno-slp-vectorize -c "$T/attention.hip" -o "$T/attention.o" 2> /dev/null
OBJS=$(ls "$CSRC"/build/*.o | grep -v "/attention.o")
mkdir -p "$ROOT/ltx-2-mlx_amd/lib/ab"
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS "$T/attention.o" -o "$ROOT/ltx-2-mlx_amd/lib/ab/broken_wait.so"
rm -rf "$T"
echo "$ROOT/ltx-2-mlx_amd/lib/ab/broken_wait.so"

#!/bin/bash
# Device assembly of the library's kernels for tools/kernel_resources.py / tools/isa_stats.py:  tools/build_asm.sh [out.s] [extra hipcc flags]
out=${1:-/tmp/isa/so3proj.s}; shift
mkdir -p "$(dirname "$out")"
exec /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -munsafe-fp-atomics -fno-slp-vectorize --cuda-device-only -S -o "$out" \
     "$(dirname "$0")/../poseestimation_amd/csrc/so3proj.hip" "$@"

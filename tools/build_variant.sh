#!/bin/bash
# usage: build_variant.sh name [-DFLAG=...]...   -> build/variants/libso3proj_<name>.so (gfx950), for A/B runs on the GPU box
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build/variants
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -munsafe-fp-atomics -fno-slp-vectorize "$@" \
    -o build/variants/libso3proj_$name.so poseestimation_amd/csrc/so3proj.hip 2>&1 | grep -v "hip-link" || true
ls -la build/variants/libso3proj_$name.so

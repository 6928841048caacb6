#!/bin/bash
# usage (here): tools/mkvariant.sh <name> [extra hipcc flags ...] — the working tree's library as build_ab/<name>.so (for tools/ab.sh on the GPU box)
R=/root/repo; name=$1; shift; mkdir -p $R/build_ab
cd $R/circom-2-arithc_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-inline-asm -mllvm -amdgpu-atomic-optimizer-strategy=None "$@" -shared -o $R/build_ab/$name.so c2a_api.hip 2>&1 | grep -E "error" ; ls -la $R/build_ab/$name.so

#!/bin/bash
# Diagnostic build of the library with in-kernel cycle stamps in fps_rows_kernel (GB_FPS_STAMPS): per-phase cycles of
# every wave go to a buffer of their own (tools/fps_stamps.py reads it).  CPU-side: builds tools/bin/libgraspbal_stamps.so
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/bin/stamps
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -munsafe-fp-atomics -fno-fast-math -Wno-unused-function"
/opt/rocm/bin/hipcc $F -fno-slp-vectorize -mllvm -amdgpu-promote-alloca-to-vector-limit=2048 -DGB_FPS_STAMPS=1 -c graspbalance_amd/csrc/fps.hip -o tools/bin/stamps/fps.o
/opt/rocm/bin/hipcc $F -c graspbalance_amd/csrc/capi.hip -o tools/bin/stamps/capi.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/libgraspbal_stamps.so tools/bin/stamps/fps.o tools/bin/stamps/capi.o
echo built tools/bin/libgraspbal_stamps.so

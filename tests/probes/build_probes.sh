#!/bin/bash
# builds the standalone measurement probes into build/ (needs dreamvla_amd/libdvla_hip.so: __graft_entry__.build() first)
set -e
cd "$(dirname "$0")/../.."
mkdir -p build
for p in gemm_probe store_probe load_probe; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 tests/probes/$p.cpp -o build/$p -Ldreamvla_amd -ldvla_hip -Wl,-rpath,'$ORIGIN/../dreamvla_amd'
done

#!/bin/sh
# measurement build of the library with the tree kernels' phase stamps compiled in (never shipped: gpurun_out/ is scratch)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared \
    -DAZG_TREE_TIMING -o alphazero_general_amd/lib/libazg_timing.so alphazero_general_amd/csrc/azg_engine.hip

#!/bin/sh
# measurement builds of the library (never shipped as the product: AZG_LIB_PATH selects them):
#   build_timing.sh tree   -> phase stamps of the tree kernels        (tools/time_tree.py)
#   build_timing.sh tower  -> per-layer / per-phase stamps of k_tower2 (tools/tower_stamps.py, tools/wide_search_phases.py)
#   build_timing.sh tuning -> the product kernels + the AZG_TOWER_BOARDS / AZG_TOWER_PSPLIT environment overrides
#                             (tools/sweep_small.py, tools/sweep_tower.py); writes libazg_tuning.so
cd "$(dirname "$0")/.." || exit 1
OUT=libazg_timing.so
case "${1:-tree}" in tower) D=-DAZG_TOWER_TIMING ;; tuning) D=-DAZG_TUNING; OUT=libazg_tuning.so ;; *) D=-DAZG_TREE_TIMING ;; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared $D \
    -o alphazero_general_amd/lib/$OUT alphazero_general_amd/csrc/azg_engine.hip

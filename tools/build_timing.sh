#!/bin/sh
# measurement builds of the library (never shipped as the product: AZG_LIB_PATH selects them) -- alphazero_general_amd/build.py owns
# the flags and the source stamp:
#   build_timing.sh tree   -> phase stamps of the tree kernels        (tools/time_tree.py)
#   build_timing.sh tower  -> per-layer / per-phase stamps of k_tower2 (tools/tower_stamps.py, tools/wide_search_phases.py)
#   build_timing.sh tuning -> the product kernels + the AZG_TOWER_BOARDS / AZG_TOWER_PSPLIT environment overrides
#                             (tools/sweep_small.py, tools/sweep_tower.py); writes libazg_tuning.so
#   build_timing.sh debug  -> -DAZG_DEBUG_BOUNDS: bounds-checked tree kernels (tools/debug_soak.py); writes libazg_debug.so
cd "$(dirname "$0")/.." || exit 1
case "${1:-tree}" in tower) V=timing-tower ;; tuning) V=tuning ;; debug) V=debug ;; *) V=timing-tree ;; esac
python -m alphazero_general_amd.build --force --variant $V

#!/bin/bash
# the profiling / A-B builds round_end.sh looks for (lib/var_*.so travel to the GPU box with the snapshot; the product never loads them)
cd "$(dirname "$0")/.." || exit 1
python profiles/build_variant.py coopstat -DYCGE_DBG_COOPSTAT=1 | tail -1
python profiles/build_variant.py batchstat -DYCGE_DBG_BATCHSTAT=1 | tail -1
python profiles/build_variant.py voxstat -DYCGE_DBG_VOXSTAT=1 | tail -1
python profiles/build_variant.py nowalkphase -DYCGE_WALK_PHASE=0 | tail -1
python profiles/build_variant.py walk1 -DYCGE_WALK_PHASE=1 | tail -1

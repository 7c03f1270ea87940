#!/bin/bash
# per-kernel times of the topologies that take the round-5 forms, with and without them (same box)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r05c
mkdir -p $OUT
cd $ROOT
python tools/topology_prof.py ${TOPOS:-conv1_same vgg_same_3x3 conv2_7x7} > $OUT/topology_prof_new.txt 2>&1
python tools/topology_prof.py ${TOPOS:-conv1_same vgg_same_3x3 conv2_7x7} --diag no_fsame,no_ring > $OUT/topology_prof_old.txt 2>&1
grep -v amdgpu.ids $OUT/topology_prof_new.txt; echo ======; grep -v amdgpu.ids $OUT/topology_prof_old.txt

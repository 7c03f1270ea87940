#!/bin/bash
# FS form of conv_x3_ws_kernel: one tile per group (shipped) against two (-DISS_WS_FS_G2 build of cnn_ws_f / cnn_ws_g: round 5's), same box,
# interleaved; 20 min of rows, both nets
cd ${GRAFT_REPO_ROOT:-$PWD}
export ISS_PREC_GUARD=0
for v in "" xfsg2 "" xfsg2; do
  lib=${v:+$PWD/inaspeechsegmenter_amd/libiss_hip_$v.so}
  echo "=== ${v:-shipped (one tile per group)}"
  ISS_LIB=$lib python tools/topology_prof.py conv1_same vgg_same_3x3 conv1_same_nopool conv1_same_conv2_same conv1_same3x3_avg 2>&1 | grep -E "^## |fs>" | sed 's/; .*//'
done

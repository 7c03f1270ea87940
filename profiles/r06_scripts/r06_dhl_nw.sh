#!/bin/bash
# conv_dhl_kernel with 8 waves (two per SIMD, shipped) against the 4-wave build (-DISS_DHL_NW=4), same box, bench geometry
cd ${GRAFT_REPO_ROOT:-$PWD}
export ISS_PREC_GUARD=0
for v in "" xnw4 "" xnw4; do
  lib=${v:+$PWD/inaspeechsegmenter_amd/libiss_hip_$v.so}
  echo "=== ${v:-shipped (8 waves)}"
  ISS_LIB=$lib python tools/topology_prof.py standin --minutes 60 2>&1 | grep -E "conv_dhl_kernel"
done

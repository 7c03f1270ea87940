#!/bin/bash
# timing-only experiment builds of conv_x3_wq3_kernel (libiss_hip_x<bits>.so, -DISS_WQ3_EXP=<bits>; wrong results on purpose):
# what the footprint fetch / the conversion / the weight DMA / the stores each cost, same box, 20-minute recording
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
L=$ROOT/inaspeechsegmenter_amd
SPECS="base f32:ISS_DIAG=no_hl"
for x in $L/libiss_hip_x*.so; do [ -e "$x" ] || continue; t=$(basename $x .so); t=${t#libiss_hip_}; SPECS="$SPECS $t:ISS_LIB=$x"; done
AB_ARGS="${AB_ARGS:---minutes 20}" bash tools/ab_env.sh segmenter $SPECS

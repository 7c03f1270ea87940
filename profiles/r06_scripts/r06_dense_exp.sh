#!/bin/bash
# what bounds the segmenter nets' first dense layer (conv_x3_pw_kernel, K = 4992 / 8320)?  An EXPERIMENTS build of the library
# (ISS_DBG=4: every activation load of that kernel from one 64 KB region = L2 hits, wrong results on purpose) against the normal path, same box
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
AB_ARGS="--minutes 20" bash tools/ab_env.sh segmenter base:ISS_LIB=$ROOT/inaspeechsegmenter_amd/libiss_hip_xexp.so l2a:ISS_LIB=$ROOT/inaspeechsegmenter_amd/libiss_hip_xexp.so,ISS_DBG=4,ISS_PREC_GUARD=0

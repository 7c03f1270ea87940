#!/bin/bash
# timing-only builds of conv_dhl_kernel (ISS_DHL_EXP bits: 4 no staging stores, 8 fragments read once, 16 no global loads, 32 no barrier):
# the dense layer's launch time per build, stand-in nets, 60 min of rows (bench geometry), same box.  The variant libraries are built by hand:
#   cd inaspeechsegmenter_amd/csrc; hipcc <CXXFLAGS of the Makefile> -DISS_DHL_NW=4 -DISS_DHL_EXP=<bits> -c cnn_dhl.hip -o /tmp/x.o;
#   hipcc --offload-arch=gfx950 -shared -fPIC -o ../libiss_hip_xd<bits>.so <every other .o> /tmp/x.o -ldl
cd ${GRAFT_REPO_ROOT:-$PWD}
export ISS_PREC_GUARD=0
for v in "" 4 8 16 32 12 28 60 ""; do
  lib=${v:+$PWD/inaspeechsegmenter_amd/libiss_hip_xd$v.so}
  echo "=== ISS_DHL_EXP=${v:-0}"
  ISS_LIB=$lib python tools/topology_prof.py standin --minutes 60 2>&1 | grep -E "conv_dhl_kernel"
done

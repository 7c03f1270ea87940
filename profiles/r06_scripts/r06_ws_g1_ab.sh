cd $GRAFT_REPO_ROOT
export ISS_PREC_GUARD=0
for lib in "" inaspeechsegmenter_amd/libiss_hip_xg1.so ""  inaspeechsegmenter_amd/libiss_hip_xg1.so; do
  echo "=== lib=${lib:-default}"
  ISS_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} python tools/topology_prof.py conv2_same conv2_3x3 ch48_96 2>&1 | grep -E "^##|ws_kernel<[0-9],[0-9],(true|false),(true|false),true"
  ISS_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} python tools/topology_prof.py standin --diag no_wq 2>&1 | grep -E "^##|ws_kernel<[0-9],[0-9],(true|false),(true|false),true"
done

#!/bin/bash
# wq / wq3 kernels: correctness (GPU tests that exercise the segmenter nets), then same-box A/B against the two-waves-per-SIMD
# kernels (ISS_DIAG=no_wq) and any timing-only experiment builds present (libiss_hip_x<bits>.so)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04b
mkdir -p $OUT
cd $ROOT
[ -n "$SKIP_TESTS" ] || { timeout 600 python -m pytest tests/test_gpu_cnn.py tests/test_gpu_segmenter.py tests/test_gpu_topologies.py -m gpu -x -q > $OUT/pytest_wq.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_wq.log; }
tail -5 $OUT/pytest_wq.log
L=$ROOT/inaspeechsegmenter_amd
SPECS="wq old:ISS_DIAG=no_wq"
for x in $L/libiss_hip_x*.so; do [ -e "$x" ] || continue; t=$(basename $x .so); t=${t#libiss_hip_}; SPECS="$SPECS $t:ISS_LIB=$x"; done
AB_ARGS="${AB_ARGS:---minutes 20}" bash tools/ab_env.sh segmenter $SPECS

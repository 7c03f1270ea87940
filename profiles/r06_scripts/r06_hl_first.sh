#!/bin/bash
# first GPU run of the CHL hand-over (conv_x3_wq3h_kernel): the tests that exercise the segmenter nets, then a same-box A/B
# against f32 NHWC between the layers (ISS_DIAG=no_hl) on the 20-minute recording
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r06_hl
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_cnn.py -m gpu -x -q -k "one_wave or exact_f32 or shared" > $OUT/pytest_hl.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_hl.log
tail -15 $OUT/pytest_hl.log
AB_ARGS="${AB_ARGS:---minutes 20}" bash tools/ab_env.sh segmenter hl f32:ISS_DIAG=no_hl

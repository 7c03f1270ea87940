#!/bin/bash
# the conv / segmenter GPU tests (incl. the precision guard), then the CHL A/B on the 20-minute recording
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r06_cnn
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests/test_gpu_cnn.py tests/test_gpu_segmenter.py -m gpu -x -q -s > $OUT/pytest_cnn.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_cnn.log
grep -E "stand-in:|inflated:|passed|failed|rc=" $OUT/pytest_cnn.log | tail -12
AB_ARGS="${AB_ARGS:---minutes 20}" bash tools/ab_env.sh segmenter hl f32:ISS_DIAG=no_hl

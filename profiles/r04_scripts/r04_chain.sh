#!/bin/bash
# chained expansion + reduction (conv_x3_pwc_kernel): vbx GPU tests, then same-box A/B against the two launches it replaces
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04j
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_vbx.py -m gpu -x -q > $OUT/pytest_vbx.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_vbx.log
tail -6 $OUT/pytest_vbx.log
[ -n "$SKIP_AB" ] || bash tools/ab_env.sh vbx chain two:ISS_DIAG=no_chain

#!/bin/bash
# soak 2: 110 randomly drawn nets (third seed base) on a 2 900-frame recording (window lists 4x longer than the suite's)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r06soak22
mkdir -p $OUT
cd $ROOT
ISS_FUZZ_NNETS=110 ISS_FUZZ_BASE=50000 ISS_FUZZ_T=2900 timeout 900 python -m pytest tests/test_gpu_fuzz_topologies.py -m gpu -q -s > $OUT/pytest_fuzz_soak2.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_fuzz_soak2.log
grep -E "passed|failed|rror|^E " $OUT/pytest_fuzz_soak2.log | cut -c1-400 | tail -30

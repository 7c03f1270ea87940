#!/bin/bash
# randomised topologies against the oracle (tests/test_gpu_fuzz_topologies.py)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r05h
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests/test_gpu_fuzz_topologies.py -m gpu -q -s > $OUT/pytest_fuzz.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_fuzz.log
grep -E "^net |overlapping|passed|failed|rror" $OUT/pytest_fuzz.log | cut -c1-220 | tail -70

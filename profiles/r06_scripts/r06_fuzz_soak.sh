#!/bin/bash
# soak on the round-6 tree (CHL, fp16 halves in the shared kernels, FS one tile per group, avg-pool on the pooled forms): 320 randomly drawn nets, a new seed base, through tests/test_gpu_fuzz_topologies.py
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r06soak
mkdir -p $OUT
cd $ROOT
ISS_FUZZ_NNETS=320 ISS_FUZZ_BASE=40000 timeout 900 python -m pytest tests/test_gpu_fuzz_topologies.py -m gpu -q -s > $OUT/pytest_fuzz_soak.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_fuzz_soak.log
grep -E "passed|failed|rror|^E " $OUT/pytest_fuzz_soak.log | cut -c1-400 | tail -30
grep -h "kernels \[" $OUT/pytest_fuzz_soak.log | tr -d "[]'" | sed 's/ *kernels //' | tr ',' '\n' | sed 's/^ *//' | sort | uniq -c | sort -rn > $OUT/fuzz_soak_kernel_histogram.txt
wc -l $OUT/fuzz_soak_kernel_histogram.txt

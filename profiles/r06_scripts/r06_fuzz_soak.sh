#!/bin/bash
# soak on the round-6 tree (CHL, fp16 halves in the shared kernels, FS one tile per group, avg-pool on the pooled forms): 320 randomly drawn nets, a new seed base, through tests/test_gpu_fuzz_topologies.py
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r06soak
mkdir -p $OUT
cd $ROOT
ISS_FUZZ_NNETS=320 ISS_FUZZ_BASE=40000 timeout 900 python -m pytest tests/test_gpu_fuzz_topologies.py -m gpu -q -s > $OUT/pytest_fuzz_soak.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_fuzz_soak.log
grep -E "passed|failed|rror|^E " $OUT/pytest_fuzz_soak.log | cut -c1-400 | tail -30
python - <<PY
import re, collections
c = collections.Counter()
for line in open('$OUT/pytest_fuzz_soak.log'):
    if 'kernels [' in line:
        for m in re.finditer(r"'([^']+)'", line[line.index('kernels ['):]): c[m.group(1)] += 1
open('$OUT/fuzz_soak_kernel_histogram.txt', 'w').write(''.join(f'{n:6d} {k}\n' for k, n in c.most_common()))
PY
wc -l $OUT/fuzz_soak_kernel_histogram.txt

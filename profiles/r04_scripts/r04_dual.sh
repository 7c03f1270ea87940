#!/bin/bash
# two-source GEMM for the projection blocks of the ResNet-101: vbx GPU tests, then same-box A/B against the two launches it replaces
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04g
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_vbx.py -m gpu -x -q > $OUT/pytest_vbx.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_vbx.log
tail -6 $OUT/pytest_vbx.log
bash tools/ab_env.sh vbx dual two:ISS_DIAG=no_dual
# the stand-alone workload entries of bench.py (companions of the default line): smoke
for w in batch archive; do
  timeout 300 python bench.py --workload $w --steps 1 --warmup 1 --files-per-gpu 8 --no-cpu-baseline > $OUT/standalone_$w.json 2> $OUT/standalone_$w.err; echo "$w rc=$? $(head -c 150 $OUT/standalone_$w.json)"
done

#!/bin/bash
# per-layer HBM counters of the ResNet-101 program (one pass of ~520 windows): FETCH_SIZE and WRITE_SIZE in separate passes
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04f
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p_$C
  timeout 400 rocprofv3 --pmc $C --kernel-trace -d /tmp/p_$C -o r -- python $ROOT/tools/layer_prof.py --minutes 2.1 --reps 1 > $OUT/layer_prof_$C.txt 2>&1
  python $ROOT/tools/pmc_by_order.py $(find /tmp/p_$C -name '*.db' | head -1) 512 2 > $OUT/vbx_by_order_$C.md 2>$OUT/by_order_$C.err
done
head -3 $OUT/layer_prof_FETCH_SIZE.txt
sed -n 1,12p $OUT/vbx_by_order_FETCH_SIZE.md

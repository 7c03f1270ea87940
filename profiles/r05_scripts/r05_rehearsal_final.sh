#!/bin/bash
# final tree: the N = 2 branch of bench.py on the one leased GPU (both ranks on device 0, --comm gloo, own launcher)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r05reh
mkdir -p $OUT
cd $ROOT
( time timeout 300 python bench.py --gpus 2 --comm gloo --steps 5 --warmup 2 > $OUT/r05_bench_2rank_rehearsal.json 2> $OUT/bench2.err ) 2> $OUT/bench2.time
echo "2-rank rc=$? $(head -c 300 $OUT/r05_bench_2rank_rehearsal.json)"; tail -5 $OUT/bench2.err; tail -3 $OUT/bench2.time

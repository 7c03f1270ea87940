#!/bin/bash
# round-5 first GPU call: the N = 2 branch of bench.py on the ONE leased GPU (both ranks on device 0, --comm gloo), started
# WITHOUT a launcher (bench.py spawns its own ranks); plus the RCCL single-rank test and the driver's N = 1 command.
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r05a
mkdir -p $OUT
cd $ROOT
( time timeout 900 python bench.py --gpus 2 --comm gloo --steps 5 --warmup 2 > $OUT/r05_bench_2rank_rehearsal.json 2> $OUT/bench2.err ) 2> $OUT/bench2.time
echo "2-rank rc=$? $(head -c 300 $OUT/r05_bench_2rank_rehearsal.json)"; tail -5 $OUT/bench2.err; tail -3 $OUT/bench2.time
timeout 600 python -m pytest tests/test_sharding.py tests/test_archive.py -q > $OUT/pytest_sharding.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_sharding.log; tail -3 $OUT/pytest_sharding.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r05_bench_first.json 2> $OUT/bench1.err ) 2> $OUT/bench1.time
echo "1-rank rc=$? $(head -c 300 $OUT/r05_bench_first.json)"; tail -3 $OUT/bench1.time

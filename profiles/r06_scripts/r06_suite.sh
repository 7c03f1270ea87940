#!/bin/bash
# GPU suite + the driver's bench command (run on the GPU box: gpurun --timeout 2400 -- 'bash profiles/r06_scripts/r06_suite.sh [tag]')
ROOT=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-mid}
OUT=$ROOT/gpurun_out/r06_$TAG
mkdir -p $OUT
cd $ROOT
( time timeout 2000 python -m pytest tests -m gpu -x -q > $OUT/r06_pytest_gpu_$TAG.log 2>&1 ) 2> $OUT/pytest.time; echo "pytest rc=$?" >> $OUT/r06_pytest_gpu_$TAG.log
tail -4 $OUT/r06_pytest_gpu_$TAG.log; tail -3 $OUT/pytest.time
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r06_bench_$TAG.json 2> $OUT/bench.err ) 2> $OUT/bench.time
python - <<PY
import json
j = json.load(open("$OUT/r06_bench_$TAG.json"))
print(j["value"], j["unit"], j["ms_per_step"], "ms/step", j["roofline"]["frac"], {k["kernel"][8:40]: round(k["ms_per_step"], 2) for k in j["roofline"]["kernels"]})
print("parity", j.get("parity_check"))
PY

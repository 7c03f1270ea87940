#!/bin/bash
# round-4 first GPU call: GPU test suite + the default bench line (new layout) + kernel stats of a short run
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04a
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
( time timeout 900 python bench.py --steps 10 --warmup 2 > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_default.time
tail -3 $OUT/bench_default.time; tail -c 1500 $OUT/bench_default.err
python - <<PY
import json
try:
    j = json.load(open("$OUT/bench_default.json"))
    r = j["roofline"]
    print("value", j["value"], "ms", j["ms_per_step"], "dominant", r["kernel"], r["frac"], r["avg_launch_ms"], "traffic", r["traffic"])
    for k in r["kernels"]: print("   ", k["kernel"], round(k["ms_per_step"],2), k["launches"], round(k["frac"],4))
    print("cpu", {k: v for k, v in j["cpu_baseline"].items() if k != "legs"})
    print("legs", json.dumps(j["cpu_baseline"]["legs"])[:3000])
    print("parity", {k: v for k, v in j["parity_check"].items() if k != "what"})
    for k, v in j.get("companions", {}).items(): print("companion", k, v.get("value"), v.get("ms_per_step"), v.get("x_realtime"), v.get("config", {}).get("ms_per_file_per_gpu"))
except Exception as e:
    print("bench parse FAILED", e)
PY

#!/bin/bash
# whole GPU suite, smoke(), the driver's bench command, the 2-rank rehearsal (self-spawned, both ranks on the one GPU, --comm gloo)
R=r05
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${R}d
mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests -m gpu -q > $OUT/${R}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/${R}_pytest_gpu.log
tail -4 $OUT/${R}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${R}_bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_default.time
tail -3 $OUT/bench_default.time
python - <<PY
import json
j = json.load(open("$OUT/${R}_bench_default.json"))
r = j["roofline"]
print("value", j["value"], "ms", j["ms_per_step"], "dominant", r["kernel"], round(r["frac"], 4), r["avg_launch_ms"], "traffic", r["traffic"])
for k in r["kernels"]: print("   ", k["kernel"], round(k["ms_per_step"],2), k["launches"], round(k["frac"],4))
c = j["cpu_baseline"]
print("cpu", c["x_realtime"], "full path all cores", c["legs"].get("full_path_all_cores"))
print("parity", {k: v for k, v in j["parity_check"].items() if k not in ("what", "classes_present")})
for k, v in j.get("companions", {}).items(): print("companion", k, v.get("value"), v.get("ms_per_step"), v.get("x_realtime"))
PY
( time timeout 900 python bench.py --gpus 2 --comm gloo --steps 5 --warmup 2 > $OUT/${R}_bench_2rank_rehearsal.json 2> $OUT/bench2.err ) 2> $OUT/bench2.time
echo "2-rank rc=$? lines=$(wc -l < $OUT/${R}_bench_2rank_rehearsal.json) $(head -c 200 $OUT/${R}_bench_2rank_rehearsal.json)"; tail -3 $OUT/bench2.time

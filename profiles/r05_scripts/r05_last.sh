#!/bin/bash
# last call of the round: whole GPU suite on the final tree, smoke(), the topology sweep (the lowering changed after r05_full.sh ran)
R=r05
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${R}z
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q > $OUT/${R}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/${R}_pytest_gpu.log
tail -3 $OUT/${R}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
python tests/topology_sweep.py --out $OUT/${R}_topology_sweep.json > $OUT/sweep.log 2>&1; tail -1 $OUT/sweep.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${R}_bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_default.time; tail -3 $OUT/bench_default.time
python - <<PY
import json
for ln in open("$OUT/sweep.log"):
    if ln.startswith('{'):
        e = json.loads(ln)
        print(f"{e['topology']:22s} {e['cnn_stage_hours_per_s']:6.2f} h/s  {e['tflops_algorithmic']:6.1f} TF  dp {e['max_abs_dprob']:.1e}")
j = json.load(open("$OUT/${R}_bench_default.json"))
r = j["roofline"]
print("value", j["value"], "ms", j["ms_per_step"], "frac", r["frac"], "traffic", r["traffic"], r.get("traffic_scale_to_this_run"), "f32", j["precision_f32"]["value"], j["precision_f32"]["frac"])
PY

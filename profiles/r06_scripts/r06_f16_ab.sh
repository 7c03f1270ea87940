#!/bin/bash
# same-box A/B of the operand-half types: split bf16 (default) against split fp16 (ISS_PREC_F16X3), 20-minute recording
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
mkdir -p gpurun_out/ab_env
for rep in 1 2; do for pr in bf16x3 f16x3; do
  timeout 300 python bench.py --minutes 20 --steps 4 --no-cpu-baseline --no-f32-companion --no-companions --precision $pr > gpurun_out/ab_env/prec_${pr}_$rep.json 2> gpurun_out/ab_env/prec_${pr}_$rep.err
  python - <<PY
import json
try:
    j = json.load(open("gpurun_out/ab_env/prec_${pr}_$rep.json")); r = j["roofline"]
    print("$pr", $rep, round(j["ms_per_step"], 2), "ms/step", j.get("precision_guard", {}).get("vad"), {k["kernel"].split(" /")[0][8:]: round(k["ms_per_step"], 2) for k in r["kernels"]})
except Exception as e:
    print("$pr", $rep, "FAILED", e)
PY
done; done

#!/bin/bash
# exact-f32 form of the weight-stationary kernel: parity, then the f32 companion of the bench with and without it (same box)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r05g
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_cnn.py -m gpu -x -q -s -k "exact_f32 or layer_semantics or ina_like" > $OUT/pytest_f32.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_f32.log
grep -E "f32 ws vs|passed|failed|Error|assert" $OUT/pytest_f32.log | tail -12
for D in "" no_f32ws; do
  ISS_DIAG=$D timeout 600 python bench.py --precision f32 --steps 3 --warmup 1 --no-cpu-baseline --no-companions > $OUT/bench_f32_${D:-new}.json 2> $OUT/bench_f32_${D:-new}.err
  python - <<PY
import json
j = json.load(open("$OUT/bench_f32_${D:-new}.json"))
r = j["roofline"]
print("${D:-new}", "value", round(j["value"], 3), "ms", round(j["ms_per_step"], 1), "frac all", round(r["all_gemm_launches"]["frac"], 3))
for k in r["kernels"][:5]: print("   ", k["kernel"], round(k["ms_per_step"], 2), k["launches"], round(k["frac"], 4))
PY
done

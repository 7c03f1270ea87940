#!/bin/bash
# Same-box A/B of two builds of the library on the default bench workload (run on the GPU box through gpurun):
#   bash tools/ab.sh tagA libA tagB libB      lib = path relative to the repo root, or "-" for the default build
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
run() {
  tag=$1; lib=$2; rep=$3
  if [ "$lib" = "-" ]; then unset ISS_LIB; else export ISS_LIB=$R/$lib; fi
  timeout 200 python $R/bench.py --no-cpu-baseline --no-f32-companion --steps 4 > $R/gpurun_out/ab_${tag}_$rep.json 2> $R/gpurun_out/ab_${tag}_$rep.err
  python - <<PY
import json
j = json.load(open("$R/gpurun_out/ab_${tag}_$rep.json"))
print("$tag", $rep, round(j["ms_per_step"], 2), {k["kernel"][8:]: (round(k["ms_per_step"], 2), k["launches"]) for k in j["roofline"]["kernels"]})
PY
}
for rep in 1 2; do run $1 $2 $rep; run $3 $4 $rep; done

#!/bin/bash
# Same-box A/B of kernel-selection switches (ISS_DIAG=no_pws+no_pws2 ... -> iss_set_diag, see include/iss.h; '+' separates
# names here because ',' separates variables) or of two builds (ISS_LIB=<other .so>) on one bench workload, run on the GPU box
# through gpurun; every variant is run twice, interleaved:
#   bash tools/ab_env.sh segmenter|vbx  tagA[:ENV=VAL[,ENV=VAL]]  tagB[:ENV=VAL...]  ...
#   e.g.  gpurun --timeout 400 -- 'bash tools/ab_env.sh vbx new old:ISS_DIAG=no_pws'
R=${GRAFT_REPO_ROOT:-$PWD}
W=$1; shift
OUT=$R/gpurun_out/ab_env
mkdir -p $OUT
run() {
  spec=$1; rep=$2
  tag=${spec%%:*}
  envs=""
  [ "$spec" != "$tag" ] && envs=$(echo "${spec#*:}" | tr ',' ' ')
  if [ "$W" = vbx ]; then args="--workload vbx --steps 2 --warmup 1 --no-cpu-baseline"; else args="--no-cpu-baseline --no-f32-companion --no-companions --timing-only --steps 4 ${AB_ARGS}"; fi
  timeout 200 env $envs python $R/bench.py $args > $OUT/${W}_${tag}_$rep.json 2> $OUT/${W}_${tag}_$rep.err
  python - <<PY
import json
try:
    j = json.load(open("$OUT/${W}_${tag}_$rep.json"))
    r = j["roofline"]
    print("$W $tag", $rep, round(j["ms_per_step"], 2), "ms/step", round(j.get("x_realtime", j.get("x_realtime_per_gpu", 0))), "x RT | other",
          round(r.get("other_kernels_ms_per_step", 0.0), 2), {k["kernel"].split(" /")[0][8:]: (round(k["ms_per_step"], 2), k["launches"]) for k in r["kernels"]})
except Exception as e:
    print("$W $tag", $rep, "FAILED", e)
PY
}
for rep in 1 2; do for spec in "$@"; do run "$spec" $rep; done; done

#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/exp6
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_cnn.py tests/test_gpu_segmenter.py -m gpu -x -q -k "shared_first or musanmix or ina_like or chunked" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
seg() { tag=$1; shift; timeout 300 env "$@" python bench.py --no-cpu-baseline --no-f32-companion --steps 4 > $OUT/seg_$tag.json 2> $OUT/seg_$tag.err
  python - <<PY
import json
try:
    j = json.load(open("$OUT/seg_$tag.json"))
    print("seg $tag", round(j["ms_per_step"], 2), "other", round(j["roofline"]["other_kernels_ms_per_step"], 2), {k["kernel"][8:18]: (round(k["ms_per_step"], 2), k["launches"]) for k in j["roofline"]["kernels"]})
except Exception as e:
    print("seg $tag FAILED", e)
PY
}
seg rows X=1
seg raw ISS_NO_FLROWS=1
seg rows2 X=1
seg raw2 ISS_NO_FLROWS=1

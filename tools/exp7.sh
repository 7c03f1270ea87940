#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/exp7
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_cnn.py tests/test_gpu_segmenter.py -m gpu -x -q -k "not pointwise and not topolog" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
timeout 300 python bench.py --no-cpu-baseline --no-f32-companion --steps 4 > $OUT/seg.json 2> $OUT/seg.err
python - <<PY
import json
j = json.load(open("$OUT/seg.json"))
print("seg", round(j["ms_per_step"], 2), "other", round(j["roofline"]["other_kernels_ms_per_step"], 2), {k["kernel"][8:18]: (round(k["ms_per_step"], 2), k["launches"]) for k in j["roofline"]["kernels"]})
PY

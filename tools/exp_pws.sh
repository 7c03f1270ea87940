#!/bin/bash
# GPU-box experiment: the streaming pointwise kernel (conv_pw.h) against the round-2 kernel (ISS_NO_PWS=1).
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/pws
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
vbx() { tag=$1; shift; timeout 300 env "$@" python bench.py --workload vbx --steps 2 --warmup 1 --no-cpu-baseline $EXTRA > $OUT/vbx_$tag.json 2> $OUT/vbx_$tag.err
  python - <<PY
import json
try:
    j = json.load(open("$OUT/vbx_$tag.json"))
    print("vbx $tag", round(j["x_realtime"]), "x RT", round(j["ms_per_step"], 1), "ms/step  xvec", round(j["config"]["xvectors_ms_per_step"], 1), "conv_ms", round(j["roofline"]["kernel_ms_per_step"], 1))
except Exception as e:
    print("vbx $tag FAILED", e)
PY
}
EXTRA=""
vbx new X=1
vbx old ISS_NO_PWS=1
for mb in 512 1024 2048 4096; do EXTRA="--workspace-mb $mb" vbx new_ws$mb X=1; done
EXTRA=""
seg() { tag=$1; shift; timeout 300 env "$@" python bench.py --no-cpu-baseline --no-f32-companion --steps 4 > $OUT/seg_$tag.json 2> $OUT/seg_$tag.err
  python - <<PY
import json
try:
    j = json.load(open("$OUT/seg_$tag.json"))
    print("seg $tag", round(j["ms_per_step"], 2), {k["kernel"][8:]: (round(k["ms_per_step"], 2), k["launches"]) for k in j["roofline"]["kernels"]}, j.get("parity_check", {}).get("segments_equal"))
except Exception as e:
    print("seg $tag FAILED", e)
PY
}
seg new X=1
seg old ISS_NO_PWS=1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d /tmp/p_vbxl -o r -- python $ROOT/bench.py --workload vbx --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
python $ROOT/tools/layer_times.py $(find /tmp/p_vbxl -name '*.db' | head -1) > $OUT/vbx_layer_times.md 2>&1
tail -32 $OUT/vbx_layer_times.md

#!/bin/bash
# vbx (configs[4]): windows per pass against tile alignment -- the library default (12 GiB of workspace = 1208 windows per pass) and caps that
# give 904 (every stage's 512-row tile count lands just under a multiple of 256 CUs), 680 and 456 windows per pass; same box
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
for rep in 1 2; do for mb in 0 9156 6888 4620; do
  python bench.py --workload vbx --steps 2 --warmup 1 --no-cpu-baseline --workspace-mb $mb > gpurun_out/vbx_ws_$mb.json 2> gpurun_out/vbx_ws_$mb.err
  python - <<PY
import json
j = json.load(open("gpurun_out/vbx_ws_$mb.json"))
print("workspace-mb $mb rep $rep:", round(j["ms_per_step"], 1), "ms per audio-hour,", round(j.get("x_realtime", 0)), "x RT")
PY
done; done

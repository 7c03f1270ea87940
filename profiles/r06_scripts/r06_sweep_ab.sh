#!/bin/bash
# same-box A/B of the CHL hand-over on topologies whose second convolution is NOT conv_x3_wq_kernel (their pooled output goes through the
# shared epilogue, conv_common.h chl_store): default against ISS_DIAG=no_hl (f32 NHWC between the layers)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
ROWS="conv2_same conv2_3x3 conv2_5x5 conv2_7x7 conv1_same vgg_same_3x3 conv1_same3x3_avg ch32_64 ch48_96 conv2_stride2 conv1_pool overlap_pool_avg conv2_4x5_same"
python tests/topology_sweep.py --only $ROWS --out gpurun_out/sweep_ab_hl.json > gpurun_out/sweep_ab_hl.log 2>&1
ISS_DIAG=no_hl python tests/topology_sweep.py --only $ROWS --out gpurun_out/sweep_ab_nohl.json > gpurun_out/sweep_ab_nohl.log 2>&1
python - <<PY
import json
a = {r['topology']: r for r in json.load(open('gpurun_out/sweep_ab_hl.json'))['results']}
b = {r['topology']: r for r in json.load(open('gpurun_out/sweep_ab_nohl.json'))['results']}
for k in a:
    print(f"{k:28s} chl {a[k]['cnn_stage_hours_per_s']:.2f}  f32 {b[k]['cnn_stage_hours_per_s']:.2f}  {100 * (a[k]['cnn_stage_hours_per_s'] / b[k]['cnn_stage_hours_per_s'] - 1):+.1f} %")
PY

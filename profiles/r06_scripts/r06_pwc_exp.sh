#!/bin/bash
# timing-only builds of conv_x3_pwc_kernel (conv_pwc.h ISS_PWC_EXP bits: 1 no MFMAs, 2 no stores, 8 no residual loads, 16 no weight
# loads): per-instantiation time of one x-vector step, same box
cd ${GRAFT_REPO_ROOT:-$PWD}
export ISS_PREC_GUARD=0
for v in "" 1 2 8 16 24 26 ""; do
  lib=${v:+$PWD/inaspeechsegmenter_amd/libiss_hip_xp$v.so}
  echo "=== ISS_PWC_EXP=${v:-0}"
  ISS_LIB=$lib python bench.py --workload vbx --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['ms_per_step'],1), {k['kernel']: round(k['ms_per_step'],1) for k in d['roofline']['kernels'] if 'pwc' in k['kernel']})"
done

#!/bin/bash
# where do the waves of the dense kernel (conv_dhl_kernel) wait?  SQ counters of one 20-minute step
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --minutes 20 --steps 1 --warmup 0 --no-cpu-baseline --no-f32-companion --no-companions"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE --kernel-trace -d /tmp/p_d -o r -- $B > /dev/null 2>&1
python $ROOT/tools/pmc_summary.py $(find /tmp/p_d -name '*.db' | head -1) > $ROOT/gpurun_out/dense_pmc.json
python - <<PY
import json
j = json.load(open("$ROOT/gpurun_out/dense_pmc.json"))
for k, v in j.items():
    if 'dhl' in k or 'wq3h' in k or 'wq_kernel' in k:
        wc = v.get('SQ_WAVE_CYCLES', 1)
        print(k[:60], 'avg us', round(v['avg_duration_us'], 1), {c: round(v.get(c, 0) / wc, 3) for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_LDS', 'SQ_ACTIVE_INST_VMEM')},
              'mfma busy', round(v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (v.get('GRBM_GUI_ACTIVE', 1) / 8 * 256 * 4), 3))
PY

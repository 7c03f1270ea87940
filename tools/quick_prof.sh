#!/bin/bash
# Run on the GPU box: kernel-time table of one bench run + two SQ counter passes on a 20-minute recording.
#   gpurun --timeout 900 -- 'bash tools/quick_prof.sh tag [ENV=VAL ...]'
TAG=${1:-q}; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/p_stats_$TAG -o r -- python $ROOT/bench.py --no-cpu-baseline --no-companions --steps 3 --warmup 1 > $OUT/bench_under_rocprof.json 2>/dev/null
python $ROOT/tools/rocprof_summary.py $(find /tmp/p_stats_$TAG -name '*.db' | head -1) $OUT/kernel_stats.md "python bench.py --no-cpu-baseline --steps 3 --warmup 1 ($*)" > /dev/null
B="python $ROOT/bench.py --minutes 20 --steps 1 --warmup 0 --no-cpu-baseline --no-companions --no-f32-companion"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace -d /tmp/p_s_$TAG -o r -- $B > /dev/null 2>&1
python $ROOT/tools/pmc_summary.py $(find /tmp/p_s_$TAG -name '*.db' | head -1) > $OUT/pmc_sq.json
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM --kernel-trace -d /tmp/p_q1_$TAG -o r -- $B > /dev/null 2>&1
python $ROOT/tools/pmc_summary.py $(find /tmp/p_q1_$TAG -name '*.db' | head -1) > $OUT/pmc_q1.json
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_VALU_MFMA_COEXEC_CYCLES SQ_IFETCH SQ_LDS_CMD_FIFO_FULL --kernel-trace -d /tmp/p_q2_$TAG -o r -- $B > /dev/null 2>&1
python $ROOT/tools/pmc_summary.py $(find /tmp/p_q2_$TAG -name '*.db' | head -1) > $OUT/pmc_q2.json
cat $OUT/kernel_stats.md | head -14
python - <<PY
import json
sq=json.load(open('$OUT/pmc_sq.json')); q1=json.load(open('$OUT/pmc_q1.json')); q2=json.load(open('$OUT/pmc_q2.json'))
for k,s in sq.items():
    if 'conv_x3' not in k: continue
    cyc=s['GRBM_GUI_ACTIVE']/8.0
    a=q1.get(k,{}); b=q2.get(k,{})
    wc=a.get('SQ_WAVE_CYCLES',1)
    mf=b.get('SQ_INSTS_MFMA',1)
    print(f"{k[:70]:70s} n={s['launches']:3d} {s['avg_duration_us']:8.1f}us clk {cyc/(s['avg_duration_us']*1e3):.2f} mfma_busy {100*s['SQ_VALU_MFMA_BUSY_CYCLES']/(cyc*1024):5.1f}% lds_conf {100*s['SQ_LDS_BANK_CONFLICT']/max(s['SQ_LDS_IDX_ACTIVE'],1):4.0f}% | wave: valu {100*a.get('SQ_ACTIVE_INST_VALU',0)/wc:4.1f}% stall {100*a.get('SQ_WAIT_INST_ANY',0)/wc:4.1f}% parked {100*a.get('SQ_WAIT_ANY',0)/wc:4.1f}% ldsstall {100*a.get('SQ_WAIT_INST_LDS',0)/wc:4.1f}% | per-mfma valu {(b.get('SQ_INSTS_VALU',0)-mf)/mf:4.2f} salu {b.get('SQ_INSTS_SALU',0)/mf:4.2f} lds {b.get('SQ_INSTS_LDS',0)/mf:4.2f} vmem {b.get('SQ_INSTS_VMEM',0)/mf:4.2f} vgpr {s.get('vgpr')}")
PY

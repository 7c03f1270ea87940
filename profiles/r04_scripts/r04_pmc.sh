#!/bin/bash
# PMC passes of the round (FETCH_SIZE / WRITE_SIZE / SQ in separate runs, the guide's recipe) -> gpurun_out/r04d, + one test file
R=r04
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04d
mkdir -p $OUT
cd $ROOT
timeout 300 python -m pytest tests/test_gpu_vbx.py -m gpu -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --minutes 20 --steps 1 --warmup 0 --no-cpu-baseline --no-f32-companion --no-companions"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_f -o r -- $B > $OUT/pmc_f.log 2>&1
python $ROOT/tools/pmc_summary.py $(find /tmp/p_f -name '*.db' | head -1) > $OUT/pmc_fetch.json
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p_w -o r -- $B > $OUT/pmc_w.log 2>&1
python $ROOT/tools/pmc_summary.py $(find /tmp/p_w -name '*.db' | head -1) > $OUT/pmc_write.json
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace -d /tmp/p_s -o r -- $B > $OUT/pmc_s.log 2>&1
python $ROOT/tools/pmc_summary.py $(find /tmp/p_s -name '*.db' | head -1) > $OUT/pmc_sq.json
python $ROOT/tools/pmc_report.py $OUT/pmc_fetch.json $OUT/pmc_write.json $OUT/pmc_sq.json $OUT/pmc_latest.json > $OUT/${R}_pmc.md
cat $OUT/${R}_pmc.md | head -14; tail -3 $OUT/pmc_f.log

#!/bin/bash
# PMC passes + topology sweep of the closing build (counter sets of tools/final_profiles.sh, known to work).
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --minutes 20 --steps 1 --warmup 0 --no-cpu-baseline --no-f32-companion"
timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_f -o r -- $B > /dev/null 2>&1
python $ROOT/tools/pmc_summary.py $(find /tmp/p_f -name '*.db' | head -1) > $OUT/pmc_fetch.json
timeout 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p_w -o r -- $B > /dev/null 2>&1
python $ROOT/tools/pmc_summary.py $(find /tmp/p_w -name '*.db' | head -1) > $OUT/pmc_write.json
timeout 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace -d /tmp/p_s -o r -- $B > /dev/null 2>&1
python $ROOT/tools/pmc_summary.py $(find /tmp/p_s -name '*.db' | head -1) > $OUT/pmc_sq.json
python $ROOT/tools/pmc_report.py $OUT/pmc_fetch.json $OUT/pmc_write.json $OUT/pmc_sq.json $OUT/pmc_latest.json > $OUT/pmc_table.md
cd $ROOT
timeout 300 python tests/topology_sweep.py --out $OUT/r03_topology_sweep.json > $OUT/sweep.log 2>&1
head -20 $OUT/pmc_table.md; tail -3 $OUT/sweep.log

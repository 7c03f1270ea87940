#!/bin/bash
# Run on the GPU box (gpurun): the measurement set committed under profiles/ at the end of a round.
#   gpurun --timeout 1500 -- 'bash tools/final_profiles.sh r01'
R=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $ROOT/bench.py > $OUT/${R}_bench_final.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d /tmp/p_stats -o r -- python $ROOT/bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2>/dev/null
python $ROOT/tools/rocprof_summary.py $(find /tmp/p_stats -name '*.db' | head -1) $OUT/${R}_bench_kernel_stats_final.md "python bench.py --no-cpu-baseline (default steps)" > /dev/null
B="python $ROOT/bench.py --minutes 20 --steps 1 --warmup 0 --no-cpu-baseline"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_f -o r -- $B > /dev/null 2>&1
python $ROOT/tools/pmc_summary.py $(find /tmp/p_f -name '*.db' | head -1) > $OUT/pmc_fetch.json
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p_w -o r -- $B > /dev/null 2>&1
python $ROOT/tools/pmc_summary.py $(find /tmp/p_w -name '*.db' | head -1) > $OUT/pmc_write.json
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace -d /tmp/p_s -o r -- $B > /dev/null 2>&1
python $ROOT/tools/pmc_summary.py $(find /tmp/p_s -name '*.db' | head -1) > $OUT/pmc_sq.json
python $ROOT/tools/pmc_report.py $OUT/pmc_fetch.json $OUT/pmc_write.json $OUT/pmc_sq.json $OUT/pmc_latest.json > $OUT/pmc_table.md
cd $ROOT
python bench.py --workload vbx --no-cpu-baseline > $OUT/${R}_vbx_1h.json 2> $OUT/vbx.err
python tools/batch_e2e.py > $OUT/${R}_batch_e2e.json 2> $OUT/e2e.err
tail -c 600 $OUT/${R}_bench_final.json; echo; cat $OUT/pmc_table.md; tail -c 400 $OUT/${R}_vbx_1h.json; echo; cat $OUT/${R}_batch_e2e.json

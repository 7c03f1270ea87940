#!/bin/bash
# Run on the GPU box (gpurun): the measurement set committed under profiles/ at the end of a round.
#   gpurun --timeout 1500 -- 'bash tools/final_profiles.sh r01'
R=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $ROOT/bench.py > $OUT/${R}_bench_final.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d /tmp/p_stats -o r -- python $ROOT/bench.py --no-cpu-baseline --no-f32-companion > $OUT/bench_under_rocprof.json 2>/dev/null
python $ROOT/tools/rocprof_summary.py $(find /tmp/p_stats -name '*.db' | head -1) $OUT/${R}_bench_kernel_stats_final.md "python bench.py --no-cpu-baseline --no-f32-companion (default steps)" > /dev/null
B="python $ROOT/bench.py --minutes 20 --steps 1 --warmup 0 --no-cpu-baseline --no-f32-companion"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_f -o r -- $B > /dev/null 2>&1
python $ROOT/tools/pmc_summary.py $(find /tmp/p_f -name '*.db' | head -1) > $OUT/pmc_fetch.json
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p_w -o r -- $B > /dev/null 2>&1
python $ROOT/tools/pmc_summary.py $(find /tmp/p_w -name '*.db' | head -1) > $OUT/pmc_write.json
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace -d /tmp/p_s -o r -- $B > /dev/null 2>&1
python $ROOT/tools/pmc_summary.py $(find /tmp/p_s -name '*.db' | head -1) > $OUT/pmc_sq.json
python $ROOT/tools/pmc_report.py $OUT/pmc_fetch.json $OUT/pmc_write.json $OUT/pmc_sq.json $OUT/pmc_latest.json > $OUT/pmc_table.md
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM --kernel-trace -d /tmp/p_q1 -o r -- $B > /dev/null 2>&1
python $ROOT/tools/pmc_summary.py $(find /tmp/p_q1 -name '*.db' | head -1) > $OUT/pmc_q1.json
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_VALU_MFMA_COEXEC_CYCLES SQ_IFETCH SQ_LDS_CMD_FIFO_FULL --kernel-trace -d /tmp/p_q2 -o r -- $B > /dev/null 2>&1
python $ROOT/tools/pmc_summary.py $(find /tmp/p_q2 -name '*.db' | head -1) > $OUT/pmc_q2.json
rocprofv3 --kernel-trace --stats -d /tmp/p_vbx -o r -- python $ROOT/bench.py --workload vbx --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $ROOT/tools/rocprof_summary.py $(find /tmp/p_vbx -name '*.db' | head -1) $OUT/${R}_vbx_kernel_stats.md "bench.py --workload vbx --steps 2 --warmup 1" > /dev/null
cd $ROOT
python tools/layer_prof.py > $OUT/${R}_vbx_layer_times.md 2>&1          # per-layer HIP-event times (rocprofv3's trace slows these launches ~2x on some boxes)
python tools/seg_layer_prof.py > $OUT/${R}_seg_layer_times.txt 2>&1
cd $ROOT
python bench.py --workload vbx > $OUT/${R}_vbx_1h.json 2> $OUT/vbx.err
python bench.py --workload archive --files-per-gpu 1250 --steps 2 --warmup 1 > $OUT/${R}_bench_archive_1250.json 2> $OUT/archive1250.err
python bench.py --workload batch > $OUT/${R}_bench_batch.json 2> $OUT/batch.err
python bench.py --workload archive > $OUT/${R}_bench_archive_1gpu.json 2> $OUT/archive.err
python tests/topology_sweep.py --out $OUT/${R}_topology_sweep.json > $OUT/sweep.log 2>&1
tail -c 600 $OUT/${R}_bench_final.json; echo; cat $OUT/pmc_table.md; tail -c 400 $OUT/${R}_vbx_1h.json; echo; tail -c 300 $OUT/${R}_bench_batch.json; tail -3 $OUT/sweep.log

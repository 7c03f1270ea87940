#!/bin/bash
# Round-6 measurement set (run on the GPU box: gpurun --timeout 2700 -- 'bash profiles/r06_scripts/r06_full.sh'); writes gpurun_out/r05f
R=r06
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${R}f
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null) | nproc $(nproc) | $(lscpu | grep -E 'Model name|^CPU\(s\)|Thread|Socket' | tr -s ' ' | tr '\n' ';')" > $OUT/host.txt; cat $OUT/host.txt
( time python $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${R}_bench_final.json 2> $OUT/bench.err ) 2> $OUT/bench.time; tail -3 $OUT/bench.time
rocprofv3 --kernel-trace --stats -d /tmp/p_stats -o r -- python $ROOT/bench.py --no-cpu-baseline --no-f32-companion --no-companions > $OUT/bench_under_rocprof.json 2>/dev/null
python $ROOT/tools/rocprof_summary.py $(find /tmp/p_stats -name '*.db' | head -1) $OUT/${R}_bench_kernel_stats_final.md "python bench.py --no-cpu-baseline --no-f32-companion --no-companions (default steps)" > /dev/null
# PMC passes, counters in separate runs.  rocprofv3 --pmc segfaults on the 1 h recording (profiles/r05_pmc_1h_attempt.txt): 20 minutes --
# the same pass geometry (~30 k slots per launch), fewer passes; bench.py scales the bytes per launch by the flops per launch
B="python $ROOT/bench.py --minutes 20 --steps 1 --warmup 0 --no-cpu-baseline --no-f32-companion --no-companions"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_f -o r -- $B > /dev/null 2>&1
python $ROOT/tools/pmc_summary.py $(find /tmp/p_f -name '*.db' | head -1) > $OUT/pmc_fetch.json
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p_w -o r -- $B > /dev/null 2>&1
python $ROOT/tools/pmc_summary.py $(find /tmp/p_w -name '*.db' | head -1) > $OUT/pmc_write.json
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace -d /tmp/p_s -o r -- $B > $OUT/pmc_run_line.json 2>/dev/null
python $ROOT/tools/pmc_summary.py $(find /tmp/p_s -name '*.db' | head -1) > $OUT/pmc_sq.json
python $ROOT/tools/pmc_report.py $OUT/pmc_fetch.json $OUT/pmc_write.json $OUT/pmc_sq.json $OUT/pmc_latest.json "bench.py --minutes 20 --steps 1 --warmup 0 --no-companions of the round-6 build (profiles/r06_scripts/r06_full.sh); rocprofv3 --pmc crashes on --minutes 60 (profiles/r05_pmc_1h_attempt.txt)" $OUT/pmc_run_line.json > $OUT/${R}_pmc.md
rocprofv3 --kernel-trace --stats -d /tmp/p_vbx -o r -- python $ROOT/bench.py --workload vbx --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $ROOT/tools/rocprof_summary.py $(find /tmp/p_vbx -name '*.db' | head -1) $OUT/${R}_vbx_kernel_stats.md "bench.py --workload vbx --steps 2 --warmup 1" > /dev/null
cd $ROOT
python tools/layer_prof.py > $OUT/${R}_vbx_layer_times.md 2>&1
python tools/seg_layer_prof.py > $OUT/${R}_seg_layer_times.txt 2>&1
python bench.py --workload vbx > $OUT/${R}_vbx_1h.json 2> $OUT/vbx.err
python bench.py --workload batch > $OUT/${R}_bench_batch.json 2> $OUT/batch.err
python bench.py --workload batch --dense-files > $OUT/${R}_bench_batch_dense.json 2> $OUT/batch_dense.err
python bench.py --workload archive > $OUT/${R}_bench_archive_1gpu.json 2> $OUT/archive.err
( time timeout 900 python bench.py --workload archive --files-per-gpu 1250 --steps 2 --warmup 1 > $OUT/${R}_bench_archive_1250.json 2> $OUT/archive1250.err ) 2> $OUT/archive1250.time
python tests/topology_sweep.py --out $OUT/${R}_topology_sweep.json > $OUT/sweep.log 2>&1
( time timeout 900 python bench.py --gpus 2 --comm gloo --steps 5 --warmup 2 > $OUT/${R}_bench_2rank_rehearsal.json 2> $OUT/bench2.err ) 2> $OUT/bench2.time
python bench.py --precision bf16x3 --no-cpu-baseline --no-companions > $OUT/${R}_bench_bf16x3.json 2> $OUT/bench_bf16.err
tail -c 400 $OUT/${R}_bench_final.json; echo; cat $OUT/${R}_pmc.md; tail -c 300 $OUT/${R}_vbx_1h.json; echo; tail -3 $OUT/sweep.log

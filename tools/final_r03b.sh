#!/bin/bash
# Round-3 closing measurement set after the pointwise / first-layer / patch-stats work (lean: ~10 GPU-minutes).
#   gpurun --timeout 840 -- 'bash tools/final_r03b.sh'
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/final
mkdir -p $OUT
cd $ROOT
( timeout 300 python -m pytest tests -m gpu -q 2>&1; echo "rc=$?" ) > $OUT/r03_pytest_gpu.log
tail -3 $OUT/r03_pytest_gpu.log
timeout 240 python bench.py > $OUT/r03_bench_final.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/p_stats -o r -- python $ROOT/bench.py --no-cpu-baseline --no-f32-companion > $OUT/bench_under_rocprof.json 2>/dev/null
python $ROOT/tools/rocprof_summary.py $(find /tmp/p_stats -name '*.db' | head -1) $OUT/r03_bench_kernel_stats_final.md "python bench.py --no-cpu-baseline --no-f32-companion (default steps)" > /dev/null
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/p_vbx -o r -- python $ROOT/bench.py --workload vbx --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $ROOT/tools/rocprof_summary.py $(find /tmp/p_vbx -name '*.db' | head -1) $OUT/r03_vbx_kernel_stats.md "bench.py --workload vbx --steps 2 --warmup 1" > /dev/null
cd $ROOT
timeout 100 python tools/layer_prof.py > $OUT/r03_vbx_layer_times.md 2> $OUT/layer_prof.err
timeout 60 python tools/seg_layer_prof.py > $OUT/r03_seg_layer_times.txt 2> $OUT/seg_layer_prof.err
timeout 240 python bench.py --workload vbx > $OUT/r03_vbx_1h.json 2> $OUT/vbx.err
timeout 200 python bench.py --workload batch > $OUT/r03_bench_batch.json 2> $OUT/batch.err
timeout 200 python bench.py --workload archive > $OUT/r03_bench_archive_1gpu.json 2> $OUT/archive.err
python - <<PY
import json
for f in ("r03_bench_final", "r03_vbx_1h", "r03_bench_batch", "r03_bench_archive_1gpu"):
    try:
        j = json.load(open("$OUT/" + f + ".json"))
        print(f, round(j["value"], 3), j["unit"], round(j["ms_per_step"], 1), "ms/step", j.get("x_realtime", ""), "frac", round(j["roofline"]["frac"], 4))
    except Exception as e:
        print(f, "FAILED", e)
PY

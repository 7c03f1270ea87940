#!/bin/bash
# round-4 validation set: whole GPU suite, the driver's bench command, kernel stats + PMC passes, topology sweep -> gpurun_out/r04c
# (copy what is to be judged into profiles/)
R=r04
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04c
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/${R}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/${R}_pytest_gpu.log
tail -4 $OUT/${R}_pytest_gpu.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${R}_bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_default.time
tail -3 $OUT/bench_default.time
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/p_stats -o r -- python $ROOT/bench.py --no-cpu-baseline --no-f32-companion --no-companions --steps 3 --warmup 1 > $OUT/bench_under_rocprof.json 2>/dev/null
python $ROOT/tools/rocprof_summary.py $(find /tmp/p_stats -name '*.db' | head -1) $OUT/${R}_bench_kernel_stats.md "python bench.py --no-cpu-baseline --no-f32-companion --no-companions --steps 3 --warmup 1" > /dev/null
B="python $ROOT/bench.py --minutes 20 --steps 1 --warmup 0 --no-cpu-baseline --no-f32-companion --no-companions"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_f -o r -- $B > /dev/null 2>&1
python $ROOT/tools/pmc_summary.py $(find /tmp/p_f -name '*.db' | head -1) > $OUT/pmc_fetch.json
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p_w -o r -- $B > /dev/null 2>&1
python $ROOT/tools/pmc_summary.py $(find /tmp/p_w -name '*.db' | head -1) > $OUT/pmc_write.json
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace -d /tmp/p_s -o r -- $B > /dev/null 2>&1
python $ROOT/tools/pmc_summary.py $(find /tmp/p_s -name '*.db' | head -1) > $OUT/pmc_sq.json
python $ROOT/tools/pmc_report.py $OUT/pmc_fetch.json $OUT/pmc_write.json $OUT/pmc_sq.json $OUT/pmc_latest.json > $OUT/${R}_pmc.md
cd $ROOT
timeout 600 python tests/topology_sweep.py --out $OUT/${R}_topology_sweep.json > $OUT/sweep.log 2>&1
python - <<PY
import json
j = json.load(open("$OUT/${R}_bench_default.json"))
r = j["roofline"]
print("value", j["value"], "ms", j["ms_per_step"], "dominant", r["kernel"], round(r["frac"], 4), r["avg_launch_ms"], "traffic", r["traffic"])
for k in r["kernels"]: print("   ", k["kernel"], round(k["ms_per_step"],2), k["launches"], round(k["frac"],4))
print("f32", j["precision_f32"]["value"], "ref-sem", j["config"]["reference_semantics"]["value"])
print("cpu", j["cpu_baseline"]["x_realtime"], "parity", {k: v for k, v in j["parity_check"].items() if k not in ("what", "classes_present")})
for k, v in j.get("companions", {}).items(): print("companion", k, v.get("value"), v.get("ms_per_step"), v.get("x_realtime"))
PY
cat $OUT/${R}_pmc.md | head -16; tail -18 $OUT/sweep.log

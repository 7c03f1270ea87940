#!/bin/bash
# round-5: topology parity with the ring / FS forms of the weight-stationary kernel, then the sweep (same box for every row)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r05b
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_topologies.py -m gpu -x -q -s -k "${TOPO_K:-topology_parity}" > $OUT/pytest_topo.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_topo.log
grep -E "vs the path|overlapping|passed|failed|Error|error|assert" $OUT/pytest_topo.log | tail -50
timeout 900 python tests/topology_sweep.py --out $OUT/r05_topology_sweep.json ${SWEEP_ONLY:+--only $SWEEP_ONLY} > $OUT/sweep.log 2>&1
python - <<PY
import json
for ln in open("$OUT/sweep.log"):
    if ln.startswith('{'):
        e = json.loads(ln)
        print(f"{e['topology']:18s} {e['cnn_stage_hours_per_s']:6.2f} h/s  {e['tflops_algorithmic']:6.1f} TF  dp {e['max_abs_dprob']:.1e}")
PY
tail -2 $OUT/sweep.log

#!/bin/bash
# soak: randomly drawn DAG-shaped nets (tests/graph_nets.random_graph) beyond the 16 of the default test run, GPU vs the oracle at 1e-4
cd ${GRAFT_REPO_ROOT:-$PWD}
mkdir -p gpurun_out
ISS_GRAPH_FUZZ_BASE=${1:-100} ISS_GRAPH_FUZZ_N=${2:-300} timeout 2400 python -m pytest tests/test_gpu_graphs.py -q -k random_graph -s 2>&1 | grep -E "random graph|passed|failed|Error|assert" > gpurun_out/r06_graph_soak.txt
tail -3 gpurun_out/r06_graph_soak.txt
python - <<'PY'
import re
errs = [float(m.group(1)) for m in re.finditer(r'max \|dp\| ([0-9.e+-]+)', open('gpurun_out/r06_graph_soak.txt').read())]
print(len(errs), 'nets; max |dp| max', max(errs), 'median', sorted(errs)[len(errs) // 2])
PY

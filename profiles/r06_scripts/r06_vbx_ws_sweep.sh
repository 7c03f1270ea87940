cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/wssweep
for rep in 1 2; do
for ws in 9156 10240 12288 14336 16384 18432 20480 22528 24576; do
  python bench.py --workload vbx --steps 3 --warmup 1 --no-cpu-baseline --workspace-mb $ws > gpurun_out/wssweep/vbx_${ws}_$rep.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/wssweep/vbx_${ws}_$rep.json").read().strip().splitlines()[-1])
print("ws $ws rep $rep", round(d['x_realtime'],1), round(d['ms_per_step'],1), d['roofline'].get('launches_per_step'))
PY
done; done

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/wssweep2
for rep in 1 2; do
for ws in 15310 16125 16935 17500 18432; do
  python bench.py --workload vbx --steps 3 --warmup 1 --no-cpu-baseline --workspace-mb $ws > gpurun_out/wssweep2/vbx_${ws}_$rep.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/wssweep2/vbx_${ws}_$rep.json").read().strip().splitlines()[-1])
print("ws $ws bc", int($ws/10.125)//8*8, "rep $rep", round(d['x_realtime'],1), round(d['ms_per_step'],1), d['roofline'].get('launches_per_step'))
PY
done; done

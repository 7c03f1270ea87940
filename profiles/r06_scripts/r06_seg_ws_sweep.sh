cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/segws
for rep in 1 2; do
for ws in 3072 4096 6144 8192 0; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-companion --no-companions --workspace-mb $ws > gpurun_out/segws/seg_${ws}_$rep.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/segws/seg_${ws}_$rep.json").read().strip().splitlines()[-1])
k={x['kernel']: (x['launches'], round(x['ms_per_step'],2)) for x in d['roofline']['kernels'][:4]}
print("ws $ws rep $rep", round(d['ms_per_step'],2), k)
PY
done; done

#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/exp5
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_vbx.py tests/test_gpu_cnn.py tests/test_vfs.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
timeout 200 python tools/layer_prof.py > $OUT/layers_new.md 2> $OUT/layers_new.err
cp $ROOT/scratch/layers_prev.md $OUT/layers_old.md # 2> $OUT/layers_old.err
python - <<PY
def grp(f):
    d={}
    on=False
    for l in open(f):
        if l.startswith('## grouped'): on=True; continue
        if on and l.startswith('| ') and not l.startswith('| kh') :
            c=[x.strip() for x in l.strip().strip('|').split('|')]
            d[c[0]]=(int(c[1]),float(c[2]),float(c[5]))
        if l.startswith('conv total'): print(f, l.strip())
    return d
a=grp("$OUT/layers_new.md"); b=grp("$OUT/layers_old.md")
for k,(n,us,bd) in sorted(a.items(), key=lambda kv:-kv[1][1]):
    o=b.get(k,(0,0,0))[1]
    if abs(o/max(us,1)-1) > 0.03: print(f"{k:32s} n={n:2d} pws2 {us:7.0f} pws {o:7.0f}  x{o/max(us,1):.2f}  bound {bd:6.0f}  new/bound {us/bd:.2f}")
PY
timeout 300 python bench.py --workload vbx --steps 2 --warmup 1 --no-cpu-baseline > $OUT/vbx.json 2>$OUT/vbx.err; python -c "
import json; j=json.load(open('$OUT/vbx.json')); print('vbx', round(j['x_realtime']), round(j['ms_per_step'],1))"

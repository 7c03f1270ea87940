#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/exp2
mkdir -p $OUT
cd $ROOT
timeout 200 python tools/layer_prof.py > $OUT/layers_new.md 2> $OUT/layers_new.err
ISS_NO_PWS=1 timeout 200 python tools/layer_prof.py > $OUT/layers_old.md 2> $OUT/layers_old.err
python - <<PY
import re
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
    print(f"{k:32s} n={n:2d} new {us:7.0f} old {o:7.0f}  x{o/max(us,1):.2f}  bound {bd:6.0f}  new/bound {us/bd:.2f}")
PY
for mb in 24576; do timeout 300 python bench.py --workload vbx --steps 2 --warmup 1 --no-cpu-baseline --workspace-mb $mb > $OUT/vbx_ws$mb.json 2>$OUT/vbx_ws$mb.err; python -c "
import json; j=json.load(open('$OUT/vbx_ws$mb.json')); print('vbx ws$mb', round(j['x_realtime']), round(j['ms_per_step'],1))"; done
timeout 300 python bench.py --no-cpu-baseline --no-f32-companion --steps 4 > $OUT/seg.json 2> $OUT/seg.err
python -c "
import json; j=json.load(open('$OUT/seg.json')); print('seg', round(j['ms_per_step'],2), {k['kernel'][8:]: (round(k['ms_per_step'],2), k['launches']) for k in j['roofline']['kernels']})"

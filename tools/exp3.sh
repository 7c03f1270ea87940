#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/exp3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d /tmp/p_$C -o r -- python $ROOT/tools/layer_prof.py --minutes 2.1 --reps 1 > $OUT/run_$C.log 2>&1
  python $ROOT/tools/pmc_by_order.py $(find /tmp/p_$C -name '*.db' | head -1) 512 2 > $OUT/layers_$C.md 2> $OUT/layers_$C.err
done
python - <<PY
f=[l.split('|') for l in open("$OUT/layers_FETCH_SIZE.md") if l.startswith('| ') and l[2].isdigit()]
w=[l.split('|') for l in open("$OUT/layers_WRITE_SIZE.md") if l.startswith('| ') and l[2].isdigit()]
seen=set()
for a,b in zip(f,w):
    key=tuple(x.strip() for x in a[2:8])
    if key in seen: continue
    seen.add(key)
    us=float(a[8]); rd=float(a[9]); wr=float(a[10]); fe=float(a[11])*2; wz=float(b[11])
    print(f"{' '.join(key):70s} {us:7.1f}us alg rd {rd:7.1f} wr {wr:7.1f} MB | FETCHx2 {fe:7.1f} ({fe/max(rd,1):.2f}x) WRITE {wz:7.1f} ({wz/max(wr,1):.2f}x) | {(fe+wz)/us/1e3:.2f} TB/s")
PY

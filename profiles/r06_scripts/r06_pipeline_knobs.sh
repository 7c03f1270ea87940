cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for cfg in "0 0 0" "3 0 0" "6 0 0" "8 0 0" "4 4800 0" "4 1200 0" "4 2400 64" "6 4800 64"; do
  set -- $cfg
  python bench.py --workload batch --workers $1 --batch-seconds $2 --batch-files $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch workers $1 seconds $2 files $3 rep $rep:', round(d['value'],3), 'h/s', round(d['ms_per_step'],1), 'ms')"
done; done

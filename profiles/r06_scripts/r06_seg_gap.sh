cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/seggap; mkdir -p gpurun_out/seggap
rocprofv3 --kernel-trace --output-format rocpd -d gpurun_out/seggap/tr -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-f32-companion --no-companions --timing-only > gpurun_out/seggap/bench.json 2> gpurun_out/seggap/err.txt
db=$(find gpurun_out/seggap/tr -name "*.db" | head -1)
python tools/gap_report.py $db sidekit_kernel 2>&1 | head -40 > gpurun_out/seggap/gaps.txt
cat gpurun_out/seggap/gaps.txt
python -c "import json; d=json.loads(open('gpurun_out/seggap/bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'])"
rm -rf gpurun_out/seggap/tr

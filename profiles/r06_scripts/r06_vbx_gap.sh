cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/vbxgap; mkdir -p gpurun_out/vbxgap
rocprofv3 --kernel-trace --output-format rocpd -d gpurun_out/vbxgap/tr -- python bench.py --workload vbx --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/vbxgap/bench.json 2> gpurun_out/vbxgap/err.txt
db=$(find gpurun_out/vbxgap/tr -name "*.db" | head -1)
echo "db=$db"
python tools/gap_report.py $db vbx_fbank 2>&1 | head -40 > gpurun_out/vbxgap/gaps.txt
cat gpurun_out/vbxgap/gaps.txt
tail -c 600 gpurun_out/vbxgap/bench.json
rm -rf gpurun_out/vbxgap/tr

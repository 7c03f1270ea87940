cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT}
for M in 20 60; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_f$M -o r -- python $ROOT/bench.py --minutes $M --steps 1 --warmup 0 --no-cpu-baseline --no-f32-companion --no-companions > /tmp/o$M.txt 2> /tmp/e$M.txt; echo "minutes $M rc=$?"; tail -5 /tmp/e$M.txt; ls /tmp/p_f$M 2>/dev/null | head -3
done
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_f20c -o r -- python $ROOT/bench.py --minutes 20 --steps 1 --warmup 0 --no-cpu-baseline --no-f32-companion > /tmp/o20c.txt 2> /tmp/e20c.txt; echo "minutes 20 with companions rc=$?"; tail -3 /tmp/e20c.txt

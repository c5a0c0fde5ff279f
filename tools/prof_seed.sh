export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_seed; mkdir -p $O; cd $R
(timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 15 --warmup 3 --no-c2 > $O/bench.log 2>&1
grep '^{"metric"' $O/bench.log | cut -c100-190
cat $O/kt/*/*_kernel_stats.csv | cut -c1-110 | head -9

export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_12m; mkdir -p $O; cd $R
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --rows 12500000 --steps 40 --warmup 5 --no-c2 > $O/bench.log 2>&1
grep '^{"metric"' $O/bench.log | cut -c1-900
cat $O/kt/*/*_kernel_stats.csv | cut -c1-200

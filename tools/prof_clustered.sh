#!/bin/bash
# where the time goes on clustered data (1000 anisotropic clusters, spread 0.3, 10M x 768): the bench's own figures, then
# rocprofv3 kernel stats of the int8-prefilter path alone
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_clustered; rm -rf $O; mkdir -p $O; cd $R
python3 tools/clustered_bench.py 10000000 768 0.3 2>/dev/null | grep CLUSTERED | sed 's/CLUSTERED //' > $R/gpurun_out/r04_clustered_10Mx768_before.json
CLUSTERED_PATHS=q8 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/q8 -- python3 tools/clustered_bench.py 10000000 768 0.3 > $O/q8.log 2>&1
f=$(ls -t $O/q8/*/*kernel_stats.csv | head -1); cp $f $R/gpurun_out/r04_clustered_q8_kernel_stats_before.csv
cut -d, -f1-6 $f | cut -c1-150 | head -14
cat $R/gpurun_out/r04_clustered_10Mx768_before.json
find $O -name "*.db" -delete; find $O -name "*trace.csv" -delete

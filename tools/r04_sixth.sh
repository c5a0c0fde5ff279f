#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_clustered2; rm -rf $O; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_adversarial.py -q 2>&1 | tail -3
CLUSTERED_PATHS=q8 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/q8 -- python3 tools/clustered_bench.py 10000000 768 0.3 > $O/q8.log 2>&1
f=$(ls -t $O/q8/*/*kernel_stats.csv | head -1); cp $f $R/gpurun_out/r04_clustered_q8_kernel_stats_after.csv
cut -d, -f1-8 $f | cut -c1-160 | head -6
python3 tools/clustered_bench.py 10000000 768 0.3 2>/dev/null | grep CLUSTERED | sed 's/CLUSTERED //' > gpurun_out/r04_clustered_10Mx768_after.json
CLUSTERED_PATHS=q8 python3 tools/clustered_bench.py 100000000 768 0.3 2>/dev/null | grep CLUSTERED | sed 's/CLUSTERED //' > gpurun_out/r04_clustered_100Mx768_after.json
cat gpurun_out/r04_clustered_10Mx768_after.json gpurun_out/r04_clustered_100Mx768_after.json
find $O -name "*.db" -delete; find $O -name "*trace.csv" -delete
python -m pytest tests -m gpu -q -x 2>&1 | tail -3

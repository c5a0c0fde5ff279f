#!/bin/bash
# round-3 profiles: rocprofv3 kernel stats of the default bench (headline + c2 + c3 + c5) and of the bare LM forward
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r03; mkdir -p $O; cd $R
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python3 bench.py > $O/default_bench.json 2> $O/default_bench.err
f=$(ls -t $O/bench/*/*kernel_stats.csv | head -1); cp $f $R/gpurun_out/r03_default_bench_kernel_stats.csv; cp $O/default_bench.json $R/gpurun_out/r03_default_bench.json
cut -d, -f1-5 $f | cut -c1-130 | head -14
PROBE_LEN=256 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/lm -- python3 tools/lm_only.py > $O/lm.log 2>&1
f=$(ls -t $O/lm/*/*kernel_stats.csv | head -1); cp $f $R/gpurun_out/r03_reranker_lm_640x256_kernel_stats.csv; tail -1 $O/lm.log | cut -c1-200
cut -d, -f1-5 $f | cut -c1-130 | head -8
find $O -name "*.db" -delete; find $O -name "*trace.csv" -delete

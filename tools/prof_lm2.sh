#!/bin/bash
# kernel stats of the reranker LM forward: PROBE_LEN / PROBE_PAIRS as in tools/lm_only.py; $1 = tag
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_lm_$1; mkdir -p $O; cd $R
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 tools/lm_only.py > $O/kt.log 2>&1
f=$(ls -t $O/kt/*/*kernel_stats.csv | head -1); cp $f $R/gpurun_out/lm_$1_kernel_stats.csv; cut -c1-150 $f | head -8; tail -1 $O/kt.log | cut -c1-300
find $O -name "*.db" -delete; find $O -name "*trace.csv" -delete

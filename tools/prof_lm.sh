#!/bin/bash
# kernel stats of the reranker LM forward sample (tools/lm_only.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_lm; mkdir -p $O; cd $R
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 tools/lm_only.py > $O/kt.log 2>&1
f=$(ls -t $O/kt/*/*kernel_stats.csv | head -1); cp $f $O/lm_kernel_stats.csv; cut -c1-160 $f | head -16; tail -2 $O/kt.log | cut -c1-900
find $O -name "*.db" -delete; find $O -name "*trace.csv" -size +2M -delete

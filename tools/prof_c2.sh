#!/bin/bash
# rocprofv3 kernel stats + bench line of BASELINE config 2 (1M x 768) — refreshes profiles/r02_c2_1Mx768_*
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_c2b; mkdir -p $O; cd $R
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --rows 1000000 --steps 50 --warmup 5 --no-c3 --no-c5 --no-cpu-baseline > $O/c2_bench.json 2> $O/c2_bench.err
f=$(ls -t $O/kt/*/*kernel_stats.csv | head -1); cp $f $O/c2_kernel_stats.csv; cut -c1-110 $f | head -8; cut -c1-300 $O/c2_bench.json
find $O -name "*.db" -delete; find $O -name "*trace.csv" -size +2M -delete

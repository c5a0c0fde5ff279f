#!/bin/bash
# config 2 (1M x 768) through the int8-prefilter path: bench line + rocprofv3 kernel stats (what the fixed costs of that path are at this size)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_c2q8; rm -rf $O; mkdir -p $O; cd $R
for sc in q8 mfma16; do
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$sc -- python3 bench.py --rows 1000000 --steps 50 --warmup 5 --scan $sc --no-c2 --no-c3 --no-c5 --no-cpu-baseline > $O/c2_$sc.json 2> $O/c2_$sc.err
f=$(ls -t $O/kt_$sc/*/*kernel_stats.csv | head -1); cp $f $O/c2_${sc}_kernel_stats.csv; echo "== $sc"; cut -d, -f1-4 $f | cut -c1-150 | head -9; cut -c1-220 $O/c2_$sc.json
done
find $O -name "*.db" -delete; find $O -name "*trace.csv" -size +2M -delete

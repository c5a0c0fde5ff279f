#!/bin/bash
# rocprofv3 kernel-trace stats of one python script: tools/prof_kernels.sh <name> <script.py> [env assignments...]
# (run on the GPU box; writes gpurun_out/<name>_kernel_stats.csv and prints its head)
name=$1; script=$2; shift 2
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$name; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/$script > $O/run.log 2>&1
f=$(ls -t $O/*/*kernel_stats.csv | head -1)
cp $f $R/gpurun_out/${name}_kernel_stats.csv
cut -c1-170 $f | head -16

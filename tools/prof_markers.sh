#!/bin/bash
# roctx ranges of the C-ABI entry points (RARC_ROCTX=1) in a rocprofv3 marker trace, next to the kernel trace
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_markers; mkdir -p $O; cd $R
export RARC_ROCTX=1 PROBE_ROWS=1000000 PROBE_DIM=768 PROBE_STORAGE=f16 PROBE_ITERS=3
timeout 300 rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d $O/kt -- python3 tools/gpu_scan_only.py > $O/run.log 2>&1
ls $O/kt/*/ | head; f=$(ls -t $O/kt/*/*marker*stats*.csv 2>/dev/null | head -1); [ -n "$f" ] && cut -c1-120 $f | head -12
find $O -name "*.db" -delete

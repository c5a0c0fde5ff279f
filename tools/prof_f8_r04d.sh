#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; rm -f gpurun_out/r04_f8_timeline.txt
for a in 1024 1025; do
RARC_Q8_TIMELINE=$R/gpurun_out/r04_f8_timeline.txt RARC_LIBRARY=$R/rag-arc_amd/lib/librarc_var_abl.so RARC_ALLOW_EXPERIMENT=1 RARC_Q8_ABL=$a PROBE_ITERS=1 PROBE_ROWS=50000000 PROBE_DIM=1024 PROBE_STORAGE=f8 python3 tools/gpu_scan_only.py 2>&1 | grep -E "SCAN|rror" 
done
head -70 gpurun_out/r04_f8_timeline.txt

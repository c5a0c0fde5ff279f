#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
export PROBE_ITERS=4 PROBE_ROWS=100000000 PROBE_DIM=1024 PROBE_STORAGE=f8
for i in 1 2; do
python3 tools/gpu_scan_only.py 2>/dev/null | grep SCAN | sed 's/^/(product: PF1 NG2) /'
for v in pf2 pf2eb pf2ng1 pf3ng1 pf4ng1; do
RARC_LIBRARY=$R/rag-arc_amd/lib/librarc_var_$v.so python3 tools/gpu_scan_only.py 2>/dev/null | grep SCAN | sed "s/^/($v) /"
done; done

#!/bin/bash
# HBM bytes of the query path's kernels: an own --pmc FETCH_SIZE pass over one bge-large query's forwards (FETCH_SIZE in KB; gfx950 counts
# 128-B requests at 64 B: bytes = FETCH_SIZE x 1024 x 2, MI355X_MICROARCH.md)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd "$R"; O=$R/gpurun_out/r06_encq_pmc; rm -rf "$O"; mkdir -p "$O"
PROBE_GEOS=bge-large PROBE_ITERS=4 timeout 600 rocprofv3 --pmc FETCH_SIZE -d "$O/pmc" -- python3 tools/enc_query_probe.py > "$O/pmc.log" 2>&1
python3 tools/pmc_summary.py "$O/pmc" all | grep -i "e32\|skinny" | tee "$O/pmc_summary.txt"
find "$O" -name "*.db" -delete 2>/dev/null

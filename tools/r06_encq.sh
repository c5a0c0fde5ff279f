#!/bin/bash
# query path: parity tests, latency against the tile kernels, per-kernel trace of one bge-large query
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$R"; O=$R/gpurun_out/r06_encq; mkdir -p "$O"
timeout 900 python3 -m pytest tests/test_gpu_encoder_query.py tests/test_gpu_encoder_f32.py -x -q -m gpu -s 2>&1 | grep -E "QUERY-PATH|ENC32|passed|failed|Error|error|assert" | tee "$O/pytest.txt"
timeout 600 python3 tools/enc_query_probe.py 2>&1 | grep ENCQ | tee "$O/latency.txt"
PROBE_GEOS=bge-large PROBE_ITERS=6 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt" -- python3 tools/enc_query_probe.py > "$O/kt.log" 2>&1
f=$(ls -t "$O"/kt/*/*kernel_stats.csv | head -1); cut -d, -f1-5 "$f" | cut -c1-130 | head -24 | tee "$O/kernels.txt"
find "$O" -name "*.db" -delete 2>/dev/null

#!/bin/bash
# round-6 starting point, one GPU call: the ADVICE r5 regressions' tests, then per-kernel traces of the three steps the round works on
# (single-query fp32-class forward, config 2's step, one rank's 12.5M-row shard step).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd "$R"; O=$R/gpurun_out/r06_baseline; mkdir -p "$O"
timeout 900 python3 -m pytest tests/test_gpu_batching.py tests/test_gpu_growable.py tests/test_gpu_flat_search.py -x -q -m gpu -rs > "$O/pytest.log" 2>&1; tail -3 "$O/pytest.log"
tools/enc_small_trace.sh 1 2>&1 | tee "$O/enc_small_n1.txt"
tools/prof.sh shard 12500000 2>&1 | tee "$O/shard.txt"
QUIET="--no-c3 --no-c5 --no-persist --no-ingest --no-api --no-f32 --no-wide --no-pairs --no-cpu-baseline"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt_c2" -- python3 bench.py --rows 1000000 $QUIET --verify-queries 8 > "$O/c2.json" 2> "$O/c2.err"
f=$(ls -t "$O"/kt_c2/*/*kernel_stats.csv | head -1); cut -d, -f1-5 "$f" | cut -c1-150 | head -14 | tee "$O/c2_kernels.txt"
tail -c 400 "$O/c2.json"
find "$O" -name "*.db" -delete 2>/dev/null

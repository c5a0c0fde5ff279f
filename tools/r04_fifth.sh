#!/bin/bash
mkdir -p gpurun_out
CLUSTERED_PATHS=q8 python3 tools/clustered_bench.py 10000000 768 0.3 2>/dev/null | grep CLUSTERED | sed 's/CLUSTERED //'
CLUSTERED_PATHS=q8 python3 tools/clustered_bench.py 10000000 768 0.6 2>/dev/null | grep CLUSTERED | sed 's/CLUSTERED //'

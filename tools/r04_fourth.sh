#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_flat_search.py tests/test_gpu_adversarial.py tests/test_gpu_fp8_corpus.py tests/test_gpu_q8_bound.py tests/test_gpu_fuzz_shapes.py tests/test_gpu_fullsize_properties.py -x -q 2>&1 | tail -4
python3 tools/clustered_bench.py 10000000 768 0.3 2>/dev/null | grep CLUSTERED | sed 's/CLUSTERED //' > gpurun_out/r04_clustered_10Mx768_select.json
cat gpurun_out/r04_clustered_10Mx768_select.json

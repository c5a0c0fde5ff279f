#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_checkpoint_stats.py -q -s 2>&1 | grep -E "CKPT|passed|failed|Error|assert" > gpurun_out/r04_ckpt_tests.log
python -m pytest tests/test_gpu_flat_search.py -q -k "truncated or batched_exact" 2>&1 | tail -3 >> gpurun_out/r04_ckpt_tests.log
python -m pytest tests/test_gpu_multirank.py tests/test_gpu_reference_pin.py -q 2>&1 | tail -3 >> gpurun_out/r04_ckpt_tests.log
cat gpurun_out/r04_ckpt_tests.log
bash tools/prof_clustered.sh

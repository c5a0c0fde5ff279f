#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
(RARC_FUZZ_SEEDS=260:420 timeout 1500 python3 -m pytest tests/test_gpu_fuzz_shapes.py -q 2>&1 | tail -1
RARC_FUZZ_SEEDS=56:96 timeout 1200 python3 -m pytest tests/test_gpu_fuzz_models.py -q 2>&1 | tail -1
SOAK_REPS=300 timeout 1200 python3 tools/search_soak.py 2>&1 | grep -v amdgpu.ids | tail -5) | tee $O/r04_soak3.txt

#!/bin/bash
# wider sweeps and soaks on the round's final tree (the suite itself runs seeds 0-13 / 0-7)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
(RARC_FUZZ_SEEDS=14:260 timeout 1500 python3 -m pytest tests/test_gpu_fuzz_shapes.py -q 2>&1 | tail -2
RARC_FUZZ_SEEDS=8:56 timeout 1200 python3 -m pytest tests/test_gpu_fuzz_models.py -q 2>&1 | tail -2
SOAK_REPS=300 timeout 900 python3 tools/enc_det_soak.py 2>&1 | grep -v amdgpu.ids | tail -3
SOAK_REPS=400 timeout 1200 python3 tools/search_soak.py 2>&1 | grep -v amdgpu.ids | tail -6
timeout 900 python3 tools/lm_det_soak.py 2>&1 | grep -v amdgpu.ids | tail -2) | tee $O/r04_soak.txt

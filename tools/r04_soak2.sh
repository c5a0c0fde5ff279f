#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
(SOAK_REPS=200 SOAK_SEQS=128 SOAK_LEN=256 timeout 900 python3 tools/enc_det_soak.py 2>&1 | grep -v amdgpu.ids | tail -2
SOAK_REPS=200 timeout 900 python3 tools/enc_det_soak.py 2>&1 | grep -v amdgpu.ids | tail -2
RARC_FUZZ_SEEDS=8:40 timeout 1200 python3 -m pytest tests/test_gpu_fuzz_models.py -q 2>&1 | tail -1) | tee $O/r04_soak2.txt

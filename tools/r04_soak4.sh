#!/bin/bash
# LM forward determinism at shapes the default soak does not visit: short prompts, the streaming attention (long prompts), small batches
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
(for sl in "640 200" "256 64" "64 640" "32 1024" "1280 96" "8 200"; do set -- $sl
  SOAK_SEQS=$1 SOAK_LEN=$2 SOAK_REPS=100 timeout 900 python3 tools/lm_det_soak.py 2>&1 | grep -v amdgpu.ids | tail -1
done
RARC_LM_ATTN=stream SOAK_SEQS=640 SOAK_LEN=200 SOAK_REPS=100 timeout 900 python3 tools/lm_det_soak.py 2>&1 | grep -v amdgpu.ids | tail -1
python3 -m pytest tests/test_gpu_reranker_lm.py tests/test_gpu_pipeline_c3.py -q 2>&1 | tail -1) | tee $O/r04_soak4.txt

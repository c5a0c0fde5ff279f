#!/bin/bash
# determinism of the fp32-class forward after RARC_MFMA_SETTLE: the shape and switch combination that showed random wrong rows
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
export SOAK_SEQS=128 SOAK_LEN=256
(echo "== 1 layer, 128 x 256 tokens, split attention + fused FFN1, 300 + 300 forwards"; SOAK_LAYERS=1 PROBE_REPS=300 python3 tools/enc_det_probe.py 2>&1 | grep "differ"
echo "== 4 layers"; SOAK_LAYERS=4 PROBE_REPS=100 python3 tools/enc_det_probe.py 2>&1 | grep "differ"
echo "== 48 x 192"; SOAK_SEQS=48 SOAK_LEN=192 SOAK_LAYERS=4 PROBE_REPS=100 python3 tools/enc_det_probe.py 2>&1 | grep "differ"
SOAK_REPS=300 SOAK_SEQS=128 SOAK_LEN=256 timeout 900 python3 tools/enc_det_soak.py 2>&1 | grep -v amdgpu.ids | tail -1
SOAK_REPS=300 SOAK_SEQS=48 SOAK_LEN=192 timeout 900 python3 tools/enc_det_soak.py 2>&1 | grep -v amdgpu.ids | tail -2
timeout 900 python3 tools/lm_det_soak.py 2>&1 | grep -v amdgpu.ids | tail -1) | tee gpurun_out/r04_determinism_after_settle.txt

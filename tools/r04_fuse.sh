#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
python3 -m pytest tests/test_gpu_encoder_f32.py tests/test_gpu_checkpoint_stats.py tests/test_gpu_mpnet.py tests/test_gpu_ingest.py tests/test_gpu_pipeline_c5.py tests/test_gpu_registry_encoder.py -q -s 2>&1 | grep "ENC32-FUSE\|CKPT-ENC\|passed\|failed\|Error" | cut -c1-230
for F in 1 0; do
  export RARC_E32_FUSE_GELU=$F
  echo "== RARC_E32_FUSE_GELU=$F"
  PROBE_SEQS=256 PROBE_TOKENS=32 RARC_ENC_PRECISION=fp32 PROBE_ITERS=20 python3 tools/enc_only.py 2>/dev/null | grep ENC
  PROBE_SEQS=64 PROBE_TOKENS=512 RARC_ENC_PRECISION=fp32 PROBE_ITERS=8 python3 tools/enc_only.py 2>/dev/null | grep ENC
  PROBE_SEQS=256 PROBE_TOKENS=128 RARC_ENC_PRECISION=fp32 PROBE_ITERS=8 python3 tools/enc_only.py 2>/dev/null | grep ENC
done

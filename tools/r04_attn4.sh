#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for v in var_e32o var_e32d; do
  export RARC_LIBRARY=$R/rag-arc_amd/lib/librarc_$v.so
  echo "=== $v"
  for i in 1 2 3 4 5 6; do python3 tools/attn_debug.py $v 2>&1 | grep -v amdgpu.ids | sed 's/.*per sequence: //' | tr '\n' ' '; echo; done
  PROBE_SEQS=64 PROBE_TOKENS=512 RARC_ENC_PRECISION=fp32 python3 tools/enc_only.py 2>/dev/null | grep ENC
  python3 -m pytest tests/test_gpu_encoder_f32.py tests/test_gpu_mpnet.py -q 2>&1 | tail -2
done

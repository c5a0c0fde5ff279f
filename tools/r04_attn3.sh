#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
for v in hip var_e32a var_e32b var_e32c var_e32d; do
  export RARC_LIBRARY=$R/rag-arc_amd/lib/librarc_$v.so
  echo "=== $v"
  python3 -m pytest tests/test_gpu_encoder_f32.py -q -s -k "split_attention or long_sequences or 768-2-12 or 1024-2-16 or 128-1-4" 2>&1 | grep "ENC32\|passed\|failed\|Error" | cut -c1-220
done

#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for v in var_x11 var_xA var_xB var_xO1; do
  export RARC_LIBRARY=$R/rag-arc_amd/lib/librarc_$v.so
  echo "=== $v"
  for i in 1 2 3; do python3 tools/attn_debug.py $v 2>&1 | grep -v amdgpu.ids | sed 's/.*per sequence: //' | tr '\n' ' '; echo; done
done

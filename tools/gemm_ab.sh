#!/bin/bash
# product library vs a variant library on the LM GEMM shapes (tools/gemm_seam_bench.py), alternated
cd $GRAFT_REPO_ROOT
for r in 1 2; do for v in "" $1; do
  if [ -n "$v" ]; then export RARC_LIBRARY=$PWD/rag-arc_amd/lib/librarc_var_$v.so; else unset RARC_LIBRARY; fi
  echo "== ${v:-product}"; python3 tools/gemm_seam_bench.py 2>/dev/null | cut -c1-70
done; done

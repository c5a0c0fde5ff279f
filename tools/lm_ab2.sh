#!/bin/bash
# LM forward (640 x 256 tokens) + GEMM shapes with the product library and a variant library, alternated
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for v in "" $1; do
  if [ -n "$v" ]; then export RARC_LIBRARY=$PWD/rag-arc_amd/lib/librarc_var_$v.so; else unset RARC_LIBRARY; fi
  echo "== ${v:-product}: $(PROBE_LEN=256 python3 tools/lm_only.py 2>&1 | tail -1)"
done; done
for v in "" $1; do
  if [ -n "$v" ]; then export RARC_LIBRARY=$PWD/rag-arc_amd/lib/librarc_var_$v.so; else unset RARC_LIBRARY; fi
  echo "== ${v:-product}"; python3 tools/gemm_seam_bench.py 2>/dev/null | cut -c1-66 | head -4
done

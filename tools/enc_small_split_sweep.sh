#!/bin/bash
# experiment: split-K policy of the fp32-class encoder's small-batch GEMMs (RARC_GEMM32_MIN_KT k tiles per slice, RARC_GEMM32_MAX_S slices)
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
for cfg in "8 8" "4 8" "4 16" "6 16" "8 16"; do
  set -- $cfg
  for n in 4 32; do
    echo -n "min_kt=$1 max_s=$2 seqs=$n: "
    RARC_GEMM32_MIN_KT=$1 RARC_GEMM32_MAX_S=$2 PROBE_SEQS=$n PROBE_ITERS=20 RARC_ENC_PRECISION=fp32 python3 tools/enc_only.py 2>/dev/null | grep ENC | sed 's/.*: //'
  done
done

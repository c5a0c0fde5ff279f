#!/bin/bash
# Beyond the suite, on the final tree: the seeded sweeps on seeds the suite does not run, and bit-for-bit repeatability soaks.
#   tools/soak.sh [first:last of the shape / lifecycle sweeps, default 14:134] [first:last of the model sweep, default 8:40]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$R"; O=$R/gpurun_out/soak; mkdir -p "$O"
S=${1:-14:134}; M=${2:-8:40}
{ echo "tree: $(git -C "$R" rev-parse --short HEAD 2>/dev/null || echo snapshot)  $(date -u +%FT%TZ)"
  echo "== shape sweep, seeds $S (tests/test_gpu_fuzz_shapes.py)";      RARC_FUZZ_SEEDS=$S timeout 3000 python3 -m pytest tests/test_gpu_fuzz_shapes.py -q -x 2>&1 | tail -2
  echo "== lifecycle sweep, seeds $S (tests/test_gpu_fuzz_lifecycle.py)"; RARC_FUZZ_SEEDS=$S timeout 3000 python3 -m pytest tests/test_gpu_fuzz_lifecycle.py -q -x 2>&1 | tail -2
  echo "== all-pairs cosine sweep, seeds $S (tests/test_gpu_similar_pairs.py)"; RARC_FUZZ_SEEDS=$S timeout 3000 python3 -m pytest tests/test_gpu_similar_pairs.py -q -x 2>&1 | tail -2
  echo "== model sweep, seeds $M (tests/test_gpu_fuzz_models.py)";       RARC_FUZZ_SEEDS=$M timeout 3000 python3 -m pytest tests/test_gpu_fuzz_models.py -q -x 2>&1 | tail -2
  echo "== search repeatability (tools/search_soak.py)";                 timeout 1200 python3 tools/search_soak.py 2>&1 | tail -10
  echo "== encoder repeatability (tools/enc_det_soak.py)";               timeout 1200 python3 tools/enc_det_soak.py 2>&1 | tail -4
  echo "== encoder QUERY PATH repeatability: 1 / 2 / 4 sequences of 32 tokens, 1000 forwards each under side-stream traffic"
  for n in 1 2 4; do SOAK_LEN=32 SOAK_SEQS=$n SOAK_REPS=1000 timeout 1200 python3 tools/enc_det_soak.py 2>&1 | tail -2; done
} 2>&1 | grep -v amdgpu.ids | tee "$O/soak.txt"

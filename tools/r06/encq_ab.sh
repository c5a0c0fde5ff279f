#!/bin/bash
# query path A/B: waves per workgroup and k steps per wave of the slab-writing weight stream
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd "$R"
for cfg in "0 0" "8 0" "4 8" "8 8" "8 4" "0 0"; do set -- $cfg
  echo "== RARC_E32Q_WAVES=$1 RARC_E32Q_KR=$2"
  RARC_E32Q_WAVES=$1 RARC_E32Q_KR=$2 PROBE_ITERS=100 timeout 300 python3 tools/enc_query_probe.py 2>&1 | grep "ENCQ.* 1 x 32"
done

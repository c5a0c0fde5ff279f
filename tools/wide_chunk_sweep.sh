#!/bin/bash
# experiment: the fused chunk size of rarc_search_wide (RARC_WIDE_FUSED_ROWS): ms per batch at k = 100 / 2000 + capacity used
for r in ${@:-131072 262144 524288}; do
  echo "== RARC_WIDE_FUSED_ROWS=$r"
  RARC_WIDE_FUSED_ROWS=$r python3 bench.py --rows 1000000 --no-c2 --no-c3 --no-c5 --no-persist --no-ingest --no-api --no-f32 --no-cpu-baseline --verify-queries 8 2>/dev/null > /tmp/wide_bigk.json
  python3 - <<'PY'
import json
j = json.load(open("/tmp/wide_bigk.json"))["wide"]
print(j["k100"]["ms_per_step"], j["k2000"]["ms_per_step"], j["k2000"]["candidate_capacity_per_query"], j["top_k_is_prefix_of_top_2000"])
PY
done

#!/bin/bash
# folded RMSNorm on / off on the bare LM forward and in the config-3 leg (development run)
python tools/lm_only.py 2>&1 | tail -1
RARC_LM_FUSE_NORM=0 python tools/lm_only.py 2>&1 | tail -1
PROBE_LEN=137 python tools/lm_only.py 2>&1 | tail -1
RARC_LM_FUSE_NORM=0 PROBE_LEN=137 python tools/lm_only.py 2>&1 | tail -1
python bench.py --rows 1000000 --no-c5 --no-c2 --no-cpu-baseline > gpurun_out/b_c3.json 2> gpurun_out/b_c3.err
python - <<'P'
import json
d = json.load(open("gpurun_out/b_c3.json")); c = d["c3"]
print(c["value"], c["lm_ms_per_step"], c["roofline"]["frac"], c["lm_parity_vs_oracle"]["max_abs_dlogit"], c["lm_parity_vs_oracle"]["within_tolerance"])
P

#!/bin/bash
# LM tests, resident vs streaming attention on the bare LM forward, config-3 leg of the bench (development run)
python -m pytest tests/test_gpu_reranker_lm.py -q > gpurun_out/t_lm.log 2>&1; grep -E "LM-PREFIX|passed|failed|FAILED" gpurun_out/t_lm.log | tail -12
python tools/lm_only.py 2>&1 | tail -1
RARC_LM_ATTN=stream python tools/lm_only.py 2>&1 | tail -1
PROBE_LEN=137 python tools/lm_only.py 2>&1 | tail -1
RARC_LM_ATTN=stream PROBE_LEN=137 python tools/lm_only.py 2>&1 | tail -1
python bench.py --rows 1000000 --no-c5 > gpurun_out/b_c3.json 2> gpurun_out/b_c3.err
python - <<'P'
import json
d = json.load(open("gpurun_out/b_c3.json")); c = d["c3"]
print(c["value"], c["lm_ms_per_step"], c["roofline"]["frac"], c["prompt_tokens"]["padded_tokens_per_step"], c["lm_parity_vs_oracle"])
P

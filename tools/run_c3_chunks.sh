#!/bin/bash
# config-3 leg at several LM chunk sizes (development run)
for c in 640 1280 2560; do
  python bench.py --rows 1000000 --no-c5 --no-c2 --no-cpu-baseline --c3-chunk $c > gpurun_out/b_c3_$c.json 2> gpurun_out/b_c3_$c.err
  python - <<P
import json
d = json.load(open("gpurun_out/b_c3_$c.json")); c = d["c3"]
print($c, c["value"], c["lm_ms_per_step"], c["roofline"]["frac"], c["prompt_tokens"]["padded_tokens_per_step"])
P
done

#!/bin/bash
# config 2 (1M x 768, batch 256, top-100): auto (fp16 MFMA scan) vs the int8 path with and without the hybrid first stage
for r in 1 2; do
for v in "auto:1:8" "q8:0:8" "q8:1:8" "q8:1:16" "q8:1:4"; do
  IFS=: read scan hy div <<< "$v"
  RARC_HYBRID=$hy RARC_HYBRID_DIV=$div python3 bench.py --rows 1000000 --scan $scan --steps 100 --warmup 10 --no-c3 --no-c5 --no-ingest --no-cpu-baseline --verify-queries 32 2>/dev/null | python3 -c "
import sys,json; j=json.loads(sys.stdin.read()); c=j.get('c2',{})
print('scan=$scan hybrid=$hy div=$div  headline ms/step', j['ms_per_step'], 'scan_ms', j['roofline']['scan_ms_per_pass'], 'launches', j['roofline']['launches_per_scan'], 'frac', j['roofline']['frac'], 'check', j['config']['full_size_check'])"
done; done

#!/bin/bash
# config 2 through bench.py with each scan_f16 variant library in turn, three rounds (box state drifts: compare within a round)
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for v in "" old mma_in; do
  if [ -n "$v" ]; then export RARC_LIBRARY=$PWD/rag-arc_amd/lib/librarc_var_$v.so; else unset RARC_LIBRARY; fi
  python3 bench.py --rows 1000000 --steps 200 --warmup 20 --scan mfma16 --no-c2 --no-c3 --no-c5 --no-cpu-baseline --verify-queries 8 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('${v:-product}', j['ms_per_step'], j['roofline']['scan_ms_per_pass'], j['config']['full_size_check'])"
done; done

#!/bin/bash
# the fp8 scan (100M x 1024) with the product library and variant libraries, alternated
cd $GRAFT_REPO_ROOT
for r in 1 2; do for v in "" "$@"; do
  if [ -n "$v" ]; then export RARC_LIBRARY=$PWD/rag-arc_amd/lib/librarc_var_$v.so; else unset RARC_LIBRARY; fi
  python3 bench.py --steps 10 --warmup 2 --storage f8 --dim 1024 --no-c2 --no-c3 --no-c5 --no-cpu-baseline --verify-queries 8 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('${v:-product} f8 1024', j['ms_per_step'], j['roofline']['frac'], j['config']['full_size_check']['rows_beating_kth'])"
done; done

#!/bin/bash
# the headline scan with the ablated instantiations of a -DRARC_Q8_ABLATIONS library (RARC_Q8_ABL: 1 = no pruning, 4 = no MFMA, 5 = neither):
# what each part of the kernel costs at the bench's own size and thresholds (results are wrong: no verification)
cd $GRAFT_REPO_ROOT
export RARC_LIBRARY=$PWD/rag-arc_amd/lib/librarc_var_abl.so   # built with -DRARC_EXPERIMENT -DRARC_Q8_ABLATIONS
export RARC_ALLOW_EXPERIMENT=1
for r in 1 2; do for a in 0 1 4 5; do
  export RARC_Q8_ABL=$a
  python3 bench.py --steps 10 --warmup 2 --no-c2 --no-c3 --no-c5 --no-cpu-baseline --verify-queries 0 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('ABL=$a f16 100M', j['ms_per_step'], j['roofline']['frac'], j['roofline']['avg_launch_ms'])"
done; done

#!/bin/bash
# config-2 shape through the int8 path with different cascade settings, kernel times by rocprofv3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c2probe; mkdir -p $O; cd $R
for m in 0 2 4 6; do
  export RARC_SPLIT_MIN=$m
  [ $m = 0 ] && unset RARC_SPLIT_MIN
  echo "== RARC_SPLIT_MIN=$m"
  python3 bench.py --rows 1000000 --scan q8 --steps 100 --warmup 10 --no-c3 --no-c5 --no-cpu-baseline --no-c2 --verify-queries 2 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['scan_ms_per_pass'], j['roofline']['launches_per_scan'])"
done
export RARC_SPLIT_MIN=2
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --rows 1000000 --scan q8 --steps 100 --warmup 10 --no-c3 --no-c5 --no-cpu-baseline --no-c2 --verify-queries 2 > /dev/null 2>&1
f=$(ls -t $O/kt/*/*kernel_stats.csv | head -1); cut -c1-150 $f | head -10
python3 tools/cand_stats.py 1000000 768 f16 | grep CAND

#!/bin/bash
# final-tree run of round 4: GPU suite, default bench, smoke, 100M clustered rows with the sticky candidate capacity
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
python3 -m pytest tests -m gpu -x -q > $O/r04_final_pytest.log 2>&1; echo "pytest exit $?" >> $O/r04_final_pytest.log
tail -5 $O/r04_final_pytest.log
python3 -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $O/r04_final_smoke.log 2>&1; tail -2 $O/r04_final_smoke.log
python3 bench.py > $O/r04_final_bench.json 2> $O/r04_final_bench.err; tail -c 1500 $O/r04_final_bench.json
CLUSTERED_PATHS=q8 python3 tools/clustered_bench.py 100000000 768 0.3 2>/dev/null | grep CLUSTERED | sed 's/CLUSTERED //' > $O/r04_clustered_100Mx768_sticky.json
cat $O/r04_clustered_100Mx768_sticky.json

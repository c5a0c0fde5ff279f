#!/bin/bash
# final-tree run: repeated attention check, GPU suite, smoke, default bench under rocprofv3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
for i in 1 2 3 4 5 6; do python3 tools/attn_debug.py final 2>&1 | grep -v amdgpu.ids | sed 's/.*per sequence: //' | tr '\n' ' '; echo; done
python3 -m pytest tests -m gpu -q > $O/r04_final_pytest.log 2>&1; echo "pytest exit $?" >> $O/r04_final_pytest.log
tail -4 $O/r04_final_pytest.log
python3 -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $O/r04_final_smoke.log 2>&1; tail -1 $O/r04_final_smoke.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_final -- python3 bench.py > $O/r04_final_bench.json 2> $O/r04_final_bench.err
f=$(ls -t $O/prof_final/*/*kernel_stats.csv | head -1); cp $f $O/r04_final_bench_kernel_stats.csv
find $O/prof_final -name "*.db" -delete; find $O/prof_final -name "*trace.csv" -delete
tail -c 600 $O/r04_final_bench.json

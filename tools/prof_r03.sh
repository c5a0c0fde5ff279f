#!/bin/bash
# Round-3 evidence: rocprofv3 kernel stats of the default bench and of the bare LM forward, PMC passes (own runs, --pmc only)
# for the headline scan (FETCH_SIZE -> profiles/traffic_r03.json) and for the LM forward's kernels (MFMA busy).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r03; rm -rf $O; mkdir -p $O; cd $R
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python3 bench.py > $O/default_bench.json 2> $O/default_bench.err
f=$(ls -t $O/bench/*/*kernel_stats.csv | head -1); cp $f $R/gpurun_out/r03_default_bench_kernel_stats.csv; cp $O/default_bench.json $R/gpurun_out/r03_default_bench.json
cut -d, -f1-5 $f | cut -c1-130 | head -12
PROBE_LEN=256 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/lm -- python3 tools/lm_only.py > $O/lm.log 2>&1
f=$(ls -t $O/lm/*/*kernel_stats.csv | head -1); cp $f $R/gpurun_out/r03_reranker_lm_640x256_kernel_stats.csv; tail -1 $O/lm.log | cut -c1-200
cut -d, -f1-5 $f | cut -c1-130 | head -8
pmc() { n=$1; shift; timeout 600 rocprofv3 --pmc "$@" -d $O/pmc_$n -- python3 $PROBE_SCRIPT > $O/pmc_$n.log 2>&1; python3 tools/pmc_summary.py $O/pmc_$n all > $R/gpurun_out/r03_pmc_$n.txt 2>&1; }
export PROBE_SCRIPT=tools/gpu_scan_only.py PROBE_ITERS=3 PROBE_ROWS=100000000 PROBE_DIM=768 PROBE_STORAGE=f16
pmc fetch_f16_100m FETCH_SIZE
grep -i "scan" $R/gpurun_out/r03_pmc_fetch_f16_100m.txt | head -3
export PROBE_SCRIPT=tools/lm_only.py PROBE_REPS=1
pmc sq_lm SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
grep -E "gemm256|attention_res" $R/gpurun_out/r03_pmc_sq_lm.txt | head -20
find $O -name "*.db" -delete; find $O -name "*trace.csv" -delete

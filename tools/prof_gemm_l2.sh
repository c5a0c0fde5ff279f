#!/bin/bash
# L2 (TCC) hit / miss counts of the LM forward's kernels: own --pmc pass, no trace domains
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_l2; rm -rf $O; mkdir -p $O; cd $R
export PROBE_LEN=256 PROBE_REPS=1
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $O/pmc -- python3 tools/lm_only.py > $O/pmc.log 2>&1
python3 tools/pmc_summary.py $O/pmc all > $R/gpurun_out/r03_pmc_l2_lm.txt 2>&1
grep -E "gemm256|attention_res" $R/gpurun_out/r03_pmc_l2_lm.txt | head -20
find $O -name "*.db" -delete

#!/bin/bash
# PMC pass for the fp16 MFMA scan at BASELINE config 2 (1M x 768): MFMA busy, clock, waits, LDS conflicts
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_c2; mkdir -p $O; cd $R
export PROBE_ROWS=1000000 PROBE_DIM=768 PROBE_STORAGE=f16 PROBE_ITERS=20
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/pmc_sq -- python3 tools/gpu_scan_only.py > $O/pmc_sq.log 2>&1
python3 tools/pmc_summary.py $O/pmc_sq all > $O/pmc_sq_c2.txt 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM -d $O/pmc_b -- python3 tools/gpu_scan_only.py > $O/pmc_b.log 2>&1
python3 tools/pmc_summary.py $O/pmc_b all > $O/pmc_b_c2.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 tools/gpu_scan_only.py > $O/kt.log 2>&1
f=$(ls -t $O/kt/*/*kernel_stats.csv | head -1); cut -c1-140 $f | head -8
find $O -name "*.db" -delete
grep -h "scan_f16" $O/pmc_sq_c2.txt $O/pmc_b_c2.txt

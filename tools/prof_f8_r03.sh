#!/bin/bash
# fp8 scan (100M x 1024) after the move to v_mfma_i32_16x16x64_i8: kernel durations (own pass) and SQ / GRBM counters (own --pmc pass)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_f8_r03; rm -rf $O; mkdir -p $O; cd $R
export PROBE_SCRIPT=tools/gpu_scan_only.py PROBE_ITERS=3 PROBE_ROWS=100000000 PROBE_DIM=1024 PROBE_STORAGE=f8
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 tools/gpu_scan_only.py > $O/kt.log 2>&1
f=$(ls -t $O/kt/*/*kernel_stats.csv | head -1); cp $f $R/gpurun_out/r03_f8_100Mx1024_kernel_stats.csv; cut -d, -f1-5 $f | cut -c1-110 | head -6
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d $O/pmc -- python3 tools/gpu_scan_only.py > $O/pmc.log 2>&1
python3 tools/pmc_summary.py $O/pmc all > $R/gpurun_out/r03_pmc_sq_f8_100m.txt 2>&1
grep -E "scan_q8" $R/gpurun_out/r03_pmc_sq_f8_100m.txt | head -8
find $O -name "*.db" -delete; find $O -name "*trace.csv" -delete

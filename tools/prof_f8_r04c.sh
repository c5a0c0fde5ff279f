#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_f8_r04c; rm -rf $O; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_fp8_corpus.py tests/test_gpu_q8_bound.py tests/test_gpu_reference_pin.py tests/test_gpu_full_size.py -q -x 2>&1 | tail -3
python -m pytest tests/test_gpu_flat_search.py -q -x -k "shadow or fp8 or f8" 2>&1 | tail -2
for i in 1 2; do
PROBE_ITERS=4 PROBE_ROWS=100000000 PROBE_DIM=1024 PROBE_STORAGE=f8 python3 tools/gpu_scan_only.py 2>/dev/null | grep SCAN
RARC_LIBRARY=$R/rag-arc_amd/lib/librarc_var_abl.so RARC_ALLOW_EXPERIMENT=1 PROBE_ITERS=4 PROBE_ROWS=100000000 PROBE_DIM=1024 PROBE_STORAGE=f8 python3 tools/gpu_scan_only.py 2>/dev/null | grep SCAN | sed 's/^/(old stride D+16) /'
done
export PROBE_ITERS=2 PROBE_ROWS=100000000 PROBE_DIM=1024 PROBE_STORAGE=f8
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/pmc2 -- python3 tools/gpu_scan_only.py > $O/pmc2.log 2>&1
python3 tools/pmc_summary.py $O/pmc2 all > $R/gpurun_out/r04_pmc_sq2_f8_100m_after.txt 2>&1
grep -E "scan_q8" $R/gpurun_out/r04_pmc_sq2_f8_100m_after.txt | sed 's/.*EE //'
find $O -name "*.db" -delete; find $O -name "*trace.csv" -delete

#!/bin/bash
# Round 4: where the fp8 scan's time goes.  (1) ablated instantiations of rarc_scan_q8_kernel<1024, 1> in a
# -DRARC_EXPERIMENT -DRARC_Q8_ABLATIONS library (wrong results by construction), 50M x 1024 fp8 rows, batch 256;
# (2) one PMC pass of the product kernel at 100M rows.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_f8_r04; rm -rf $O; mkdir -p $O; cd $R
OUT=$R/gpurun_out/r04_f8_attribution.txt; : > $OUT
export PROBE_ITERS=4 PROBE_ROWS=50000000 PROBE_DIM=1024 PROBE_STORAGE=f8
echo "== product library" >> $OUT
python3 tools/gpu_scan_only.py 2>/dev/null | grep SCAN >> $OUT
echo "== ablation library (RARC_Q8_ABL: 1 no pruning | 4 no LDS reads + no MFMA | 5 = 1+4 | 17 no conversion, no pruning | 21 fetch + LDS write only | 32769 converted, not written, no pruning | 32773 fetch + conversion only)" >> $OUT
for a in 0 1 4 5 17 21 32769 32773; do
  RARC_LIBRARY=$R/rag-arc_amd/lib/librarc_var_abl.so RARC_ALLOW_EXPERIMENT=1 RARC_Q8_ABL=$a python3 tools/gpu_scan_only.py 2>/dev/null | grep SCAN >> $OUT
done
cat $OUT
export PROBE_ITERS=2 PROBE_ROWS=100000000
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $O/pmc1 -- python3 tools/gpu_scan_only.py > $O/pmc1.log 2>&1
python3 tools/pmc_summary.py $O/pmc1 all > $R/gpurun_out/r04_pmc_sq_f8_100m.txt 2>&1
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/pmc2 -- python3 tools/gpu_scan_only.py > $O/pmc2.log 2>&1
python3 tools/pmc_summary.py $O/pmc2 all > $R/gpurun_out/r04_pmc_sq2_f8_100m.txt 2>&1
grep -E "scan_q8" $R/gpurun_out/r04_pmc_sq_f8_100m.txt $R/gpurun_out/r04_pmc_sq2_f8_100m.txt | cut -c1-400 | head -12
tail -3 $O/pmc1.log
find $O -name "*.db" -delete; find $O -name "*trace.csv" -delete

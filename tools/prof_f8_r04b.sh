#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
OUT=$R/gpurun_out/r04_f8_attribution.txt; : > $OUT
export PROBE_ITERS=4 PROBE_ROWS=50000000 PROBE_DIM=1024 PROBE_STORAGE=f8
echo "== product library" >> $OUT
python3 tools/gpu_scan_only.py 2>/dev/null | grep SCAN >> $OUT
echo "== ablation library (RARC_Q8_ABL: 1 no pruning | 4 no LDS reads + no MFMA | 5 = 1+4 | 9 no pruning, no threshold refresh | 17 no conversion, no pruning | 21 fetch + LDS write only | 32769 converted, not written, no pruning | 32773 fetch + conversion only)" >> $OUT
for a in 0 1 4 5 9 17 21 32769 32773; do
  RARC_LIBRARY=$R/rag-arc_amd/lib/librarc_var_abl.so RARC_ALLOW_EXPERIMENT=1 RARC_Q8_ABL=$a python3 tools/gpu_scan_only.py 2>/dev/null | grep SCAN >> $OUT
done
export PROBE_ROWS=50000000 PROBE_DIM=768 PROBE_STORAGE=f16
echo "== fp16 rows x 768 (32x32x32 MFMA chain), same library" >> $OUT
for a in 0 1 4 5; do
  RARC_LIBRARY=$R/rag-arc_amd/lib/librarc_var_abl.so RARC_ALLOW_EXPERIMENT=1 RARC_Q8_ABL=$a python3 tools/gpu_scan_only.py 2>/dev/null | grep SCAN >> $OUT
done
cat $OUT

#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests/test_gpu_fp8_corpus.py tests/test_gpu_q8_bound.py tests/test_gpu_full_size.py tests/test_gpu_fuzz_shapes.py tests/test_gpu_adversarial.py -q -x 2>&1 | tail -3
python -m pytest tests/test_gpu_flat_search.py -q -x 2>&1 | tail -2
export RARC_LIBRARY=$R/rag-arc_amd/lib/librarc_var_abl.so RARC_ALLOW_EXPERIMENT=1 PROBE_ITERS=4 PROBE_ROWS=100000000 PROBE_DIM=1024 PROBE_STORAGE=f8
for i in 1 2 3; do
RARC_Q8_ABL=512 python3 tools/gpu_scan_only.py 2>/dev/null | grep SCAN | sed 's/^/(barrier at the end) /'
RARC_Q8_ABL=0 python3 tools/gpu_scan_only.py 2>/dev/null | grep SCAN | sed 's/^/(barrier behind MFMA) /'
done
rm -f gpurun_out/r04_f8_timeline_eb.txt
RARC_Q8_TIMELINE=$R/gpurun_out/r04_f8_timeline_eb.txt RARC_Q8_ABL=1024 PROBE_ITERS=1 PROBE_ROWS=50000000 python3 tools/gpu_scan_only.py > /dev/null 2>&1
head -34 gpurun_out/r04_f8_timeline_eb.txt

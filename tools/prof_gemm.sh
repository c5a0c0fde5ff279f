#!/bin/bash
# PMC pass on the encoder GEMM shapes (ours and the vendor library's kernels side by side)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_gemm; mkdir -p $O; cd $R
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/pmc -- python3 tools/gemm_ref_torch.py > $O/pmc.log 2>&1
python3 tools/pmc_summary.py $O/pmc all 2>&1 | grep -v "elementwise\|fill\|copy" | head -40
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 tools/gemm_ref_torch.py > $O/kt.log 2>&1
f=$(ls -t $O/kt/*/*kernel_stats.csv | head -1); cut -c1-200 $f | head -14

#!/bin/bash
# Round 5, VERDICT r4 item 5: do the fp8 scan's streaming phase (8.1 ms per 50M rows) and its matrix phase (7.1 ms of MFMA issue)
# overlap or add?  Measurement build of scan_q8 (wrong results on purpose: -DRARC_EXPERIMENT), 50M x 1024 fp8 rows:
#   abl 0 full kernel | 1 no pruning | 5 no pruning, no MFMA | 65537 no pruning, HALF the MFMAs (waves 4-7 issue none) | 65536 half the MFMAs
# If half the MFMAs costs ~ (full + skeleton) / 2 the phases add (a scheduling problem); if it costs ~ the skeleton, the matrix pipe was the bound.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$R"
tools/build_variant_any.sh scan_q8 abl -DRARC_EXPERIMENT -DRARC_Q8_ABLATIONS || exit 1
export RARC_LIBRARY=$R/rag-arc_amd/lib/librarc_var_abl.so RARC_ALLOW_EXPERIMENT=1 PROBE_ROWS=${1:-50000000} PROBE_DIM=1024 PROBE_STORAGE=f8 PROBE_ITERS=3
for abl in 0 1 5 65537 65536 131073 0; do RARC_Q8_ABL=$abl python3 tools/gpu_scan_only.py 2>/dev/null | grep SCAN; done
# the same three variants under PMC (own passes, --pmc only): shader clock = GRBM_GUI_ACTIVE / kernel time, MFMA pipe busy, LDS active
if [ "${2:-}" = "pmc" ]; then
  for abl in 0 5 65537 131073; do
    rm -rf gpurun_out/prof_f8ovl/pmc_$abl
    RARC_Q8_ABL=$abl timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES -d gpurun_out/prof_f8ovl/pmc_$abl -- python3 tools/gpu_scan_only.py > gpurun_out/prof_f8ovl/pmc_$abl.log 2>&1
    echo "== abl $abl"; grep SCAN gpurun_out/prof_f8ovl/pmc_$abl.log; python3 tools/pmc_summary.py gpurun_out/prof_f8ovl/pmc_$abl all 2>/dev/null | grep -i "scan_q8" | head -8
  done
fi

#!/bin/bash
# SQ counters of the two attention kernels (their own passes): is the split-operand attention bound by VALU issue?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_attn_pmc; rm -rf $O; mkdir -p $O; cd $R
export PROBE_SEQS=64 PROBE_TOKENS=512
for P in fp32 fp16; do
  export RARC_ENC_PRECISION=$P
  timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY -d $O/sq_$P -- python3 tools/enc_only.py > $O/sq_$P.log 2>&1
  timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/g_$P -- python3 tools/enc_only.py > $O/g_$P.log 2>&1
done
python3 tools/pmc_summary.py $O/sq_fp32 rarc_e32_attention_split | tee $R/gpurun_out/r04_pmc_attention.txt
python3 tools/pmc_summary.py $O/g_fp32 rarc_e32_attention_split | tee -a $R/gpurun_out/r04_pmc_attention.txt
python3 tools/pmc_summary.py $O/sq_fp16 rarc_attention_mfma_shared | tee -a $R/gpurun_out/r04_pmc_attention.txt
python3 tools/pmc_summary.py $O/g_fp16 rarc_attention_mfma_shared | tee -a $R/gpurun_out/r04_pmc_attention.txt
find $O -name "*.db" -delete; find $O -name "*trace.csv" -delete; find $O -name "*counter_collection.csv" -size +2M -delete

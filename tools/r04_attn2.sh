#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
python3 -m pytest tests/test_gpu_encoder_f32.py tests/test_gpu_mpnet.py tests/test_gpu_checkpoint_stats.py tests/test_gpu_ingest.py tests/test_gpu_batching.py tests/test_gpu_pipeline_c5.py -q 2>&1 | tail -5
for A in split mfma32; do
  export RARC_E32_ATTN=$A
  echo "== RARC_E32_ATTN=$A"
  PROBE_SEQS=64 PROBE_TOKENS=512 RARC_ENC_PRECISION=fp32 python3 tools/enc_only.py 2>/dev/null | grep ENC
  PROBE_SEQS=256 PROBE_TOKENS=128 RARC_ENC_PRECISION=fp32 python3 tools/enc_only.py 2>/dev/null | grep ENC
done
unset RARC_E32_ATTN
export PROBE_SEQS=64 PROBE_TOKENS=512 RARC_ENC_PRECISION=fp32
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_attn -- python3 tools/enc_only.py > $O/prof_attn.log 2>&1
f=$(ls -t $O/prof_attn/*/*kernel_stats.csv | head -1); cp $f $O/r04_enc_fp32_64x512_kernel_stats.csv; cut -d, -f1-5 $f | cut -c1-50,100-180 | head -7
find $O/prof_attn -name "*.db" -delete; find $O/prof_attn -name "*trace.csv" -delete

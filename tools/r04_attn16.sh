#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
python3 -m pytest tests/test_gpu_encoder.py tests/test_gpu_mpnet.py tests/test_gpu_checkpoint_stats.py tests/test_gpu_ingest.py -q 2>&1 | tail -4
for A in shared wave; do
  export RARC_ENC_ATTN=$A
  echo "== RARC_ENC_ATTN=$A"
  PROBE_SEQS=64 PROBE_TOKENS=512 RARC_ENC_PRECISION=fp16 python3 tools/enc_only.py 2>/dev/null | grep ENC
  PROBE_SEQS=256 PROBE_TOKENS=128 RARC_ENC_PRECISION=fp16 python3 tools/enc_only.py 2>/dev/null | grep ENC
  PROBE_SEQS=256 PROBE_TOKENS=64 RARC_ENC_PRECISION=fp16 python3 tools/enc_only.py 2>/dev/null | grep ENC
done
unset RARC_ENC_ATTN
export PROBE_SEQS=64 PROBE_TOKENS=512 RARC_ENC_PRECISION=fp16
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_attn16 -- python3 tools/enc_only.py > $O/prof_attn16.log 2>&1
f=$(ls -t $O/prof_attn16/*/*kernel_stats.csv | head -1); cp $f $O/r04_enc_fp16_64x512_kernel_stats.csv
python3 - $f <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:7]:
    print("  ", r["Name"][:70].ljust(70), r["Calls"].rjust(5), f'{float(r["AverageNs"]) / 1e3:9.1f} us', r["Percentage"][:5])
PY
find $O/prof_attn16 -name "*.db" -delete; find $O/prof_attn16 -name "*trace.csv" -delete

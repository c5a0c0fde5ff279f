#!/bin/bash
# Round-2 evidence: rocprofv3 kernel stats of the bench commands + PMC passes (own runs, --pmc only) for the
# fp16 scan, the fp8 scan and the encoder.  Run through gpurun; summaries are copied to profiles/ by hand.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r02; mkdir -p $O; cd $R
kt() { # name, bench args...
  n=$1; shift
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$n -- python3 bench.py "$@" > $O/${n}_bench.json 2> $O/${n}_bench.err
  f=$(ls -t $O/kt_$n/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/${n}_kernel_stats.csv
}
kt default
kt shard12m --rows 12500000 --steps 40 --warmup 5 --no-c2 --no-c3 --no-c5
kt c2 --rows 1000000 --steps 50 --warmup 5 --no-c3 --no-c5 --no-cpu-baseline
kt f8_100m --storage f8 --dim 1024 --steps 10 --no-c5
pmc() { # name, counters..., then env/cmd via PROBE_* already exported
  n=$1; shift
  timeout 600 rocprofv3 --pmc "$@" -d $O/pmc_$n -- python3 $PROBE_SCRIPT > $O/pmc_$n.log 2>&1
  python3 tools/pmc_summary.py $O/pmc_$n all > $O/pmc_$n.txt 2>&1
}
export PROBE_SCRIPT=tools/gpu_scan_only.py PROBE_ITERS=3
export PROBE_ROWS=100000000 PROBE_DIM=768 PROBE_STORAGE=f16
pmc fetch_f16_100m FETCH_SIZE
pmc sq_f16_100m SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
export PROBE_ROWS=100000000 PROBE_DIM=1024 PROBE_STORAGE=f8
pmc fetch_f8_100m FETCH_SIZE
pmc sq_f8_100m SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
export PROBE_ROWS=12500000 PROBE_DIM=1024 PROBE_STORAGE=f8 PROBE_ITERS=6
pmc sq_f8_12m SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
export PROBE_SCRIPT=tools/enc_only.py
pmc sq_encoder SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_encoder -- python3 tools/enc_only.py > $O/kt_encoder.log 2>&1
f=$(ls -t $O/kt_encoder/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/encoder_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_lm -- python3 tools/lm_only.py > $O/kt_lm.log 2>&1
f=$(ls -t $O/kt_lm/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/lm_kernel_stats.csv
export PROBE_SEQS=32
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_enc32 -- python3 tools/enc_only.py > $O/kt_enc32.log 2>&1
f=$(ls -t $O/kt_enc32/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/encoder_32seq_kernel_stats.csv
unset PROBE_SEQS
export PROBE_SCRIPT=tools/gpu_scan_only.py PROBE_ROWS=1000000 PROBE_DIM=768 PROBE_STORAGE=f16 PROBE_ITERS=20
pmc sq_c2_1m SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
python3 tools/enc_batch_sweep.py 2>/dev/null | grep ENC > $O/encoder_batch_sweep.txt
python3 tools/gemm_k_sweep.py 2>/dev/null | grep "^M=" > $O/gemm_k_sweep.txt
python3 tools/gemm_lm_bench.py 2>/dev/null | grep "^M=" > $O/gemm_lm_shapes.txt
find $O -name "*.db" -delete; find $O -name "*.csv" -size +4M -delete
ls $O; for f in $O/*_bench.json; do echo "== $f"; cut -c1-400 $f; done; cat $O/pmc_*.txt | grep -v "^$" | head -120

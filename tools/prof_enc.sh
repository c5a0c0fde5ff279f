export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_enc; mkdir -p $O; cd $R
python3 tools/encoder_bench.py > $O/enc.log 2>&1; cat $O/enc.log | tail -8
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 tools/encoder_bench.py > $O/enc_prof.log 2>&1
cat $O/kt/*/*_kernel_stats.csv | cut -c1-150 | head -12
python3 tools/gemm_bench.py 2>&1 | tail -8

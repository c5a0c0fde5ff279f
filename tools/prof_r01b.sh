export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r01b
mkdir -p $O
cd $R
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_100m -- python3 bench.py --steps 20 --warmup 3 --no-c2 > $O/kt_100m_bench.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_1m -- python3 bench.py --rows 1000000 --steps 50 --warmup 5 --no-c2 > $O/kt_1m_bench.log 2>&1
PROBE_ROWS=100000000 PROBE_ITERS=3 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch_100m -- python3 tools/gpu_scan_only.py > $O/pmc_fetch_100m.log 2>&1
PROBE_ROWS=100000000 PROBE_ITERS=3 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write_100m -- python3 tools/gpu_scan_only.py > $O/pmc_write_100m.log 2>&1
PROBE_ROWS=100000000 PROBE_ITERS=3 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/pmc_sq_100m -- python3 tools/gpu_scan_only.py > $O/pmc_sq_100m.log 2>&1
python3 tools/pmc_summary.py $O/pmc_fetch_100m > $O/pmc_fetch_100m.txt 2>&1
python3 tools/pmc_summary.py $O/pmc_write_100m > $O/pmc_write_100m.txt 2>&1
python3 tools/pmc_summary.py $O/pmc_sq_100m > $O/pmc_sq_100m.txt 2>&1
find $O -name "*.db" -size +20M -delete
find $O -name "*kernel_stats.csv" | head; cat $O/pmc_*.txt; tail -1 $O/kt_100m_bench.log | cut -c1-600

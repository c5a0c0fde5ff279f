#!/bin/bash
# Round-4 evidence on the FINAL tree: rocprofv3 kernel stats of the default bench, the FETCH_SIZE PMC passes of both headline
# scans (own runs, --pmc only), the single-GPU shard lines (what ONE rank of an N-rank run does, collective forced) for the
# projection in DESIGN 6, the encoder's per-rank query slices.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r04; rm -rf $O; mkdir -p $O; cd $R
timeout 1800 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python3 bench.py > $O/default_bench.json 2> $O/default_bench.err
f=$(ls -t $O/bench/*/*kernel_stats.csv | head -1); cp $f $R/gpurun_out/r04_default_bench_kernel_stats.csv; cp $O/default_bench.json $R/gpurun_out/r04_default_bench.json
cut -d, -f1-5 $f | cut -c1-130 | head -14
pmc() { n=$1; shift; timeout 900 rocprofv3 --pmc "$@" -d $O/pmc_$n -- python3 tools/gpu_scan_only.py > $O/pmc_$n.log 2>&1; python3 tools/pmc_summary.py $O/pmc_$n all > $R/gpurun_out/r04_pmc_$n.txt 2>&1; }
export PROBE_ITERS=3 PROBE_ROWS=100000000 PROBE_DIM=768 PROBE_STORAGE=f16
pmc fetch_f16_100m FETCH_SIZE
export PROBE_DIM=1024 PROBE_STORAGE=f8
pmc fetch_f8_100m FETCH_SIZE
grep -i "scan" $R/gpurun_out/r04_pmc_fetch_f16_100m.txt $R/gpurun_out/r04_pmc_fetch_f8_100m.txt | head -4
find $O -name "*.db" -delete; find $O -name "*trace.csv" -delete
# one rank's share of the 100M-row corpus on 2 / 4 / 8 GPUs, exchange forced through RCCL on the one rank there is
for rows in 50000000 25000000 12500000; do
  RARC_FORCE_DIST=1 python3 bench.py --rows $rows --c5-rows $rows --no-c2 --no-c3 --no-ingest --no-persist --no-cpu-baseline --verify-queries 32 > $R/gpurun_out/r04_shard_${rows}_bench.json 2> $O/shard_$rows.err
  python3 -c "
import json; j=json.load(open('$R/gpurun_out/r04_shard_${rows}_bench.json'))
print('SHARD $rows x768 f16: ms/step', j['ms_per_step'], 'scan frac', j['roofline']['frac'], 'exchange ms', j['config']['exchange_ms_per_step'], '| x1024 f8 (config 5 leg): ms/step', j['c5']['ms_per_step'], {k: j['c5'][k] for k in j['c5'] if 'encoder' in k and 'ms' in k})"
done
python3 tools/enc_batch_sweep.py 2>/dev/null | grep ENC | sed 's/^/fp16 encoder: /' > $R/gpurun_out/r04_encoder_batch_sweep.txt
RARC_ENC_PRECISION=fp32 python3 tools/enc_batch_sweep.py 2>/dev/null | grep ENC | sed 's/^/fp32-class encoder: /' >> $R/gpurun_out/r04_encoder_batch_sweep.txt
cat $R/gpurun_out/r04_encoder_batch_sweep.txt
python3 -c "
import json; j=json.load(open('$R/gpurun_out/r04_default_bench.json'))
print('DEFAULT value', j['value'], 'ms/step', j['ms_per_step'], 'roofline', j['roofline']['frac'], j['roofline']['avg_launch_ms'])
for k in ('c2','c3','c5','persistence'):
    if k in j: print(k, {kk: vv for kk, vv in j[k].items() if not isinstance(vv, (dict, list))})
print('cpu', j.get('cpu_baseline', {}).get('value'))
for c in j.get('ingest', {}).get('configs', []): print(c['model'][:9], c['precision'], c['tokens_per_text'], c['value'], c['encoder_mfma_frac_of_2500'])
"

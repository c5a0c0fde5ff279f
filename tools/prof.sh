#!/bin/bash
# ONE entry point for the profiles under profiles/ (run it on the GPU box: `gpurun -- 'tools/prof.sh <what> [args]'`).
# Everything lands in gpurun_out/prof_<what>/ ; copy what you want judged into profiles/ and list it in profiles/INDEX.md.
#
#   tools/prof.sh bench                  rocprofv3 --kernel-trace --stats of the default `python bench.py` (the driver's command)
#   tools/prof.sh shard <rows>           one rank's shard of the headline corpus, exchange forced on the one rank (RARC_FORCE_DIST=1):
#                                        bench line + per-kernel trace (12500000 / 25000000 / 50000000 = 100M over 8 / 4 / 2 GPUs)
#   tools/prof.sh pmc <rows> <dim> <f16|f8> [counters...]   a --pmc pass of the scan alone (tools/gpu_scan_only.py; own run, no
#                                        trace options next to --pmc); default counter FETCH_SIZE -> HBM bytes per launch
#   tools/prof.sh c2                     config 2 alone (1M x 768, two pipelined search contexts): bench line + per-kernel trace
#   tools/prof.sh encq                   the encoder's query path: parity tests, latency against the tile kernels, per-kernel trace (tools/r06_encq.sh)
#   tools/prof.sh api                    the plugin-surface legs only (bench.py `api` object)
#   tools/prof.sh wide                   the wide-row leg (10M x 1536) with a kernel trace
#   tools/prof.sh pairs                  the all-pairs cosine leg (100k x 1024 entities) with a kernel trace
#   tools/prof.sh f32                    the storage="f32" leg (10M x 768 fp32 rows)
#   tools/prof.sh enc [fp32|fp16]        encoder batch sweep (tools/enc_batch_sweep.py)
#   tools/prof.sh power <out> <cmd...>   board power / clocks sampled while <cmd> runs (tools/power_sample.sh)
#   tools/prof.sh vmem                   what the HIP runtime accepts of hipMemMap (tools/vmem_probe.py)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
what=${1:-bench}; shift
O=$R/gpurun_out/prof_$what; mkdir -p "$O"; cd "$R"
QUIET="--no-c2 --no-c3 --no-c5 --no-persist --no-ingest --no-api --no-f32 --no-wide --no-pairs --no-cpu-baseline"
stats() { f=$(ls -t "$1"/*/*kernel_stats.csv "$1"/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cut -d, -f1-5 "$f" | cut -c1-140 | head -${2:-14}; }
case $what in
  bench)
    timeout 2400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt" -- python3 bench.py "$@" > "$O/bench.json" 2> "$O/bench.err"
    stats "$O/kt"; tail -c 600 "$O/bench.json";;
  shard)
    rows=${1:-12500000}; shift
    RARC_FORCE_DIST=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt_$rows" -- python3 bench.py --rows $rows $QUIET --verify-queries 32 "$@" > "$O/shard_$rows.json" 2> "$O/shard_$rows.err"
    stats "$O/kt_$rows"
    python3 -c "import json; j=json.load(open('$O/shard_$rows.json')); print('SHARD $rows: ms/step', j['ms_per_step'], 'scan ms', j['roofline']['scan_ms_per_pass'], 'scan frac', j['roofline']['frac'], 'exchange ms', j['config']['exchange_ms_per_step'])";;
  pmc)
    export PROBE_ROWS=${1:-100000000} PROBE_DIM=${2:-768} PROBE_STORAGE=${3:-f16} PROBE_ITERS=3; shift; shift; shift
    timeout 900 rocprofv3 --pmc ${@:-FETCH_SIZE} -d "$O/pmc" -- python3 tools/gpu_scan_only.py > "$O/pmc.log" 2>&1
    python3 tools/pmc_summary.py "$O/pmc" all | tee "$O/pmc_summary.txt" | grep -i scan | head -6;;
  c2)
    timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt" -- python3 bench.py --rows 1000000 --steps 400 --warmup 20 $QUIET --verify-queries 32 "$@" > "$O/c2.json" 2> "$O/c2.err"
    stats "$O/kt" 12
    python3 -c "import json; j=json.load(open('$O/c2.json')); print('C2 (profiled): ms/step', j['ms_per_step'], j['config']['full_size_check'])"
    timeout 600 python3 bench.py --rows 1000000 --steps 400 --warmup 20 $QUIET --verify-queries 32 "$@" 2>/dev/null > "$O/c2_unprofiled.json"
    python3 -c "import json; j=json.load(open('$O/c2_unprofiled.json')); print('C2 (unprofiled): ms/step', j['ms_per_step'], 'q/s', j['value'])";;
  encq)  tools/r06_encq.sh;;
  api)   timeout 900 python3 bench.py --no-c3 --no-c5 --no-persist --no-ingest --no-f32 --no-wide --no-pairs "$@" > "$O/api.json" 2> "$O/api.err"; python3 -c "import json; print(json.dumps(json.load(open('$O/api.json'))['api'], indent=1))";;
  wide)  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt" -- python3 bench.py --rows 1000000 --no-c2 --no-c3 --no-c5 --no-persist --no-ingest --no-api --no-f32 --no-pairs --no-cpu-baseline --verify-queries 8 "$@" > "$O/wide.json" 2> "$O/wide.err"; stats "$O/kt" 10; python3 -c "import json; print(json.dumps(json.load(open('$O/wide.json'))['wide']))";;
  f32)   timeout 900 python3 bench.py --rows 1000000 --no-c2 --no-c3 --no-c5 --no-persist --no-ingest --no-api --no-wide --no-pairs --no-cpu-baseline --verify-queries 8 "$@" > "$O/f32.json" 2> "$O/f32.err"; python3 -c "import json; print(json.dumps(json.load(open('$O/f32.json'))['f32']))";;
  pairs) timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt" -- python3 bench.py --rows 1000000 --no-c2 --no-c3 --no-c5 --no-persist --no-ingest --no-api --no-f32 --no-wide --verify-queries 8 "$@" > "$O/pairs.json" 2> "$O/pairs.err"; stats "$O/kt" 8; python3 -c "import json; print(json.dumps(json.load(open('$O/pairs.json'))['pairs']))";;
  enc)   RARC_ENC_PRECISION=${1:-fp32} python3 tools/enc_batch_sweep.py 2>/dev/null | grep ENC | tee "$O/enc_${1:-fp32}.txt";;
  power) out=$1; shift; tools/power_sample.sh "$O/$out" "$@";;
  vmem)  python3 tools/vmem_probe.py 2>&1 | grep -v amdgpu.ids | tee "$O/vmem_probe.txt";;
  *) echo "unknown profile: $what (see the header of tools/prof.sh)"; exit 2;;
esac
find "$O" -name "*.db" -delete 2>/dev/null

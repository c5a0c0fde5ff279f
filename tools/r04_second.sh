#!/bin/bash
# round 4, second GPU contact: ingest tests, persistence tests, the whole suite, the bench's ingest leg
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ingest.py tests/test_gpu_persistence.py -x -q > gpurun_out/r04_ingest_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r04_ingest_tests.log
python -m pytest tests -m gpu -q > gpurun_out/r04_gpu_suite.log 2>&1
echo "suite rc=$?" >> gpurun_out/r04_gpu_suite.log
python bench.py --rows 1000000 --steps 3 --no-c2 --no-c3 --no-c5 --no-cpu-baseline > gpurun_out/r04_bench_ingest.json 2> gpurun_out/r04_bench_ingest.err
tail -30 gpurun_out/r04_ingest_tests.log; tail -5 gpurun_out/r04_gpu_suite.log; python -c "
import json; r=json.load(open('gpurun_out/r04_bench_ingest.json'))
for c in r['ingest']['configs']: print({k:c[k] for k in ('model','precision','tokens_per_text','documents','value','tokens_per_s','sequences_per_call','tokenize_share','tokenizer_tokens_per_s','python_tokenizer_tokens_per_s','encoder_only_documents_per_s','encoder_mfma_frac_of_2500')})
"; tail -5 gpurun_out/r04_bench_ingest.err

#!/bin/bash
# small-batch encoder: tests, timing, kernel stats at 32x32 (rocprofv3 under timeout: it has hung at exit before)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/enc_small; mkdir -p $O
cd $R
python3 -m pytest tests/test_gpu_encoder.py -x -q -m gpu 2>&1 | tail -3
python3 tools/encoder_bench_small.py 2>&1 | grep bge
cat > /tmp/one.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, torch
from oracle import cpu_ref
from rag_arc_amd.encapsulation.embeddings.hip_bert import HipBertEncoder
sd = cpu_ref.random_bert_state_dict(1024, 24, 16, 4096, vocab=2000, max_pos=512, seed=1)
enc = HipBertEncoder(sd, num_heads=16, precision="fp16")
ids = np.random.default_rng(0).integers(1, 2000, (32, 32)).astype(np.int32)
for _ in range(6): enc.forward(ids)
torch.cuda.synchronize()
PY
cd /tmp
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt2 -- python3 /tmp/one.py > $O/kt2.log 2>&1
f=$(ls -t $O/kt2/*/*kernel_stats.csv | head -1); cat $f | cut -c1-160 | head -12

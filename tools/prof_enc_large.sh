#!/bin/bash
# per-kernel times of one bge-large forward at 256 x 32 tokens
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/enc_large; mkdir -p $O
cat > /tmp/one.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, torch
from oracle import cpu_ref
from rag_arc_amd.encapsulation.embeddings.hip_bert import HipBertEncoder
sd = cpu_ref.random_bert_state_dict(1024, 24, 16, 4096, vocab=2000, max_pos=512, seed=1)
enc = HipBertEncoder(sd, num_heads=16, precision="fp16")
ids = np.random.default_rng(0).integers(1, 2000, (256, 32)).astype(np.int32)
for _ in range(6): enc.forward(ids)
torch.cuda.synchronize()
PY
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 /tmp/one.py > $O/kt.log 2>&1
f=$(ls -t $O/kt/*/*kernel_stats.csv | head -1); cut -c1-140 $f | head -12

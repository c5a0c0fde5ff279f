#!/bin/bash
# fp16 scan with idle waves (nq < 256): parity suites that vary nq, then latency of small batches at 1M rows and the api leg
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd "$R"
timeout 1200 python3 -m pytest tests/test_gpu_flat_search.py tests/test_gpu_hybrid_scan.py tests/test_gpu_fuzz_shapes.py tests/test_gpu_adversarial.py tests/test_gpu_reference_pin.py -x -q -m gpu 2>&1 | tail -3
python3 - <<'PY'
import sys, time, os
sys.path.insert(0, os.getcwd())
import torch
from rag_arc_amd.hip import binding as B
from rag_arc_amd.hip.engine import FlatIndexF16
lib = B.load_library()
n, d = 1_000_000, 768
rows = torch.zeros((n, d), dtype=torch.float16, device="cuda")
B.check(lib.rarc_synth_rows_f16(rows.data_ptr(), d, d, 0, n, 1234, 0))
idx = FlatIndexF16(d)
idx.add_rows_f16(rows, 1.001, n_valid=n)
q = torch.zeros((256, d), dtype=torch.float32, device="cuda")
B.check(lib.rarc_synth_rows_f32(q.data_ptr(), d, d, 0, 256, 4321, 0))
full = idx.search_device(q, 100)
for nq in (1, 8, 32, 33, 64, 128, 200, 256):
    ids, sc = idx.search_device(q[:nq], 100)
    same = bool(torch.equal(ids, full[0][:nq]) and torch.equal(sc, full[1][:nq]))
    for _ in range(5): idx.search_device(q[:nq], 100)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    hs = []
    for _ in range(100):
        hs.append(idx.search_async(q[:nq], 100))
        if len(hs) > 2: hs.pop(0).result()
    for h in hs: h.result()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
    lat = []
    for _ in range(30):
        t1 = time.perf_counter(); idx.search(q[:nq], 100); lat.append(time.perf_counter() - t1)
    print(f"NQ {nq:4d}: pipelined {dt*1e3:.3f} ms per batch, sync search p50 {sorted(lat)[15]*1e3:.3f} ms, same bits as the 256-query batch: {same}")
PY

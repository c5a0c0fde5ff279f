"""resident vs streaming attention: which sequences differ, for a few geometries (development probe)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import cpu_ref as oracle
from rag_arc_amd.core.rerank import HipCausalLM

def run(H, LAYERS, NQ, NKV, DH, I, n, L, P, starts=None, pstarts=None):
    V = 1000
    sd = oracle.random_qwen3_state_dict(H, LAYERS, NQ, NKV, DH, I, vocab=V, seed=H + L)
    lm = HipCausalLM(sd, NQ, NKV, DH, rms_norm_eps=1e-6, rope_theta=1e6)
    dev = lm.device
    rng = np.random.default_rng(L + P)
    ids = rng.integers(5, V, (n, L)).astype(np.int32)
    start = np.array((starts or [0, L - 1, 7, L // 2, 33, 1, L - 40, 64])[:n], np.int32)
    for r in range(n): ids[r, :start[r]] = 0
    kw = {}
    if P:
        npre = 4
        while (npre * P) % 128: npre += 1
        pre = rng.integers(5, V, (npre, P)).astype(np.int32)
        pstart = np.array([(pstarts or (0, P - 1, 5, P // 2))[i % 4] for i in range(npre)], np.int32)
        for i in range(npre): pre[i, :pstart[i]] = 0
        print("  prefix_kv ...", flush=True)
        handle = lm.prefix_kv_device(torch.from_numpy(pre).to(dev), torch.from_numpy(pstart).to(dev))
        torch.cuda.synchronize(); print("  prefix_kv done", flush=True)
        po = [(i % (npre + 1)) - 1 for i in range(n)]
        kw = dict(prefix=handle, prefix_of=torch.tensor(po, dtype=torch.int32, device=dev))
    f = lambda: lm.yes_no_logits_device(torch.from_numpy(ids).to(dev), torch.from_numpy(start).to(dev), 11, 42, **kw).float().cpu().numpy()
    os.environ["RARC_LM_ATTN"] = "stream"; b = f(); print("  streaming done", flush=True)
    os.environ.pop("RARC_LM_ATTN", None); a = f(); print("  resident done", flush=True)
    os.environ.pop("RARC_LM_ATTN", None)
    d = np.abs(a - b).max(axis=1)
    print(f"H={H} layers={LAYERS} {NQ}/{NKV} dh={DH} n={n} L={L} P={P}: max diff per sequence {np.array2string(d, precision=4)} start={start.tolist()} prefix_of={kw and po}")

run(1024, 2, 16, 8, 128, 3072, 8, 160, 96)
run(1024, 1, 16, 8, 128, 3072, 8, 160, 96)
run(1024, 1, 16, 8, 128, 3072, 8, 160, 0)
run(1024, 1, 16, 8, 128, 3072, 8, 160, 96, pstarts=(0, 0, 0, 0))
run(1024, 1, 16, 8, 128, 3072, 8, 160, 96, starts=[0] * 8)
run(512, 1, 4, 2, 128, 512, 8, 160, 96)
run(512, 1, 8, 2, 128, 512, 8, 160, 96)
run(512, 1, 4, 4, 128, 512, 8, 160, 96)
run(1024, 1, 16, 8, 128, 3072, 8, 96, 32)

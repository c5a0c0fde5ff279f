"""Per-sequence difference between two builds of the library on a 1-layer encoder (debugging aid): run once per library
(RARC_LIBRARY), saves embeddings to gpurun_out/attn_dbg_<tag>.npy"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import cpu_ref
from rag_arc_amd.encapsulation.embeddings.hip_bert import HipBertEncoder
tag = sys.argv[1]
H, heads, I, L = 128, 2, 256, 160
sd = cpu_ref.random_bert_state_dict(H, 1, heads, I, vocab=300, max_pos=L, seed=9)
lens = np.array([160, 32, 31, 64, 63, 33, 96, 1, 128, 159, 65, 17], np.int32)
rng = np.random.default_rng(9)
ids = rng.integers(1, 300, (len(lens), L)).astype(np.int32)
for r, l in enumerate(lens):
    ids[r, l:] = 0
for Lc in (160, 32):
    sel = lens <= Lc
    e = HipBertEncoder(sd, num_heads=heads, pooling="mean", precision="fp32").forward(ids[sel][:, :Lc], lens[sel], normalize=False).cpu().numpy()
    np.save(f"gpurun_out/attn_dbg_{tag}_{Lc}.npy", e)
    w = cpu_ref.bert_forward_f32(sd, ids[sel][:, :Lc], lens[sel], heads, normalize=False, pooling="mean", dtype=np.float64)
    print(tag, Lc, "lens", lens[sel].tolist(), "max|e - f64| per sequence:", np.abs(e - w).max(axis=1).round(7).tolist())

"""Three reranker LM forwards at Qwen3-Reranker-0.6B geometry (bench.py's c3.reranker_lm sample) for rocprofv3."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
a = types.SimpleNamespace(lm_queries=int(os.environ.get("PROBE_QUERIES", 8)), k=100)
print(bench.leg_reranker_lm(torch, np, a, torch.device("cuda", 0), 0))

"""fp32-class / fp16 bge-large forward latency at small batches (1, 4, 8, 32 sequences of 32 tokens), development probe."""
import os, sys, time
sys.path.insert(0, os.getcwd())
exec(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "enc_only.py")).read().split("for _ in range(6):")[0])
for n in (32, 8, 4):
    t_ = tok[:n].contiguous(); l_ = lens[:n].contiguous()
    for _ in range(3): enc.forward_device(t_, l_)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): enc.forward_device(t_, l_)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
    print(f"ENC {os.environ.get('RARC_ENC_PRECISION', 'fp16')} {n} seqs x {L} tokens: {dt*1e3:.3f} ms")

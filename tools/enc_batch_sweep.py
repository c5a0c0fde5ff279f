import os, sys, time
sys.path.insert(0, os.getcwd())
exec(open("tools/enc_only.py").read().split("for _ in range(6):")[0])
for n in (256, 128, 64, 32):
    t_ = tok[:n].contiguous(); l_ = lens[:n].contiguous()
    for _ in range(3): enc.forward_device(t_, l_)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): enc.forward_device(t_, l_)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"ENC {n} seqs x {L} tokens: {dt*1e3:.3f} ms")

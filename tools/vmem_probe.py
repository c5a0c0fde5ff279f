#!/usr/bin/env python3
"""Probe of HIP virtual-memory behaviour behind rarc_vmem_* (csrc/vmem.hip): which sequences of piece sizes hipMemMap
accepts inside one reservation.  Every experiment runs in a FRESH process (the runtime keeps state across reservations).
Usage: python tools/vmem_probe.py            (all experiments)
       python tools/vmem_probe.py one <reserve MiB> <piece MiB> <piece MiB> ...   ('/' = destroy the arena and start a new one)"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MiB = 1 << 20


def one(reserve_mib, pieces):
    import torch

    from rag_arc_amd.hip import binding as B

    lib = B.load_library()
    torch.zeros(1, device="cuda")
    out, h, at = [], None, 0
    for p in pieces + ["/"]:
        if p == "/":
            if h is not None:
                x = torch.as_tensor(type("A", (), {"__cuda_array_interface__": {"shape": (at,), "typestr": "|u1", "data": (lib.rarc_vmem_base(h), False), "version": 2, "strides": None}})(), device="cuda") if at else None
                if x is not None:
                    x.fill_(3)
                    torch.cuda.synchronize()
                    out.append("rw-ok" if int(x[::4096].sum().item()) == 3 * len(x[::4096]) else "RW-BAD")
                    del x
                out.append(f"destroy={lib.rarc_vmem_destroy(h)}")
            h, at = None, 0
            continue
        if h is None:
            h = ctypes.c_void_p()
            B.check(lib.rarc_vmem_create(0, int(reserve_mib * MiB), 4 << 30, ctypes.byref(h)), "create")
        at += int(float(p) * MiB)
        rc = lib.rarc_vmem_grow(h, at)
        out.append(f"{p}:{'ok' if rc == 0 else 'FAIL'}")
        if rc != 0:
            at -= int(float(p) * MiB)
    print(" ".join(out))


if len(sys.argv) > 1 and sys.argv[1] == "one":
    one(float(sys.argv[2]), sys.argv[3:])
    sys.exit(0)

EXPERIMENTS = {
    "uniform 2 MiB x 24": (256, ["2"] * 24),
    "uniform 64 MiB x 8": (1024, ["64"] * 8),
    "uniform 1 GiB x 6": (8192, ["1024"] * 6),
    "uniform 3 MiB x 8 (not a power of two)": (256, ["3"] * 8),
    "buddy doubling 2,2,4,8,...,1024,1024": (8192, ["2", "2", "4", "8", "16", "32", "64", "128", "256", "512", "1024", "1024", "1024"]),
    "2 then 30": (256, ["2", "30"]),
    "2 then 8 8 8 6": (256, ["2", "8", "8", "8", "6"]),
    "small then big at a big boundary: 2 x 32 then 64 x 4": (1024, ["2"] * 32 + ["64"] * 4),
    "uniform 2 MiB, destroy, uniform 2 MiB": (256, ["2"] * 6 + ["/"] + ["2"] * 6),
    "uniform 2 MiB, destroy, uniform 8 MiB": (256, ["2"] * 6 + ["/"] + ["8"] * 6),
    "uniform 8 MiB, destroy, uniform 2 MiB": (256, ["8"] * 6 + ["/"] + ["2"] * 12),
    "one piece only, odd size 37.5 MiB": (256, ["37.5"]),
    "first odd then uniform: 5 then 2 2 2": (256, ["5", "2", "2", "2"]),
}
for name, (reserve, pieces) in EXPERIMENTS.items():
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "one", str(reserve)] + pieces, capture_output=True, text=True)
    print(f"{name}: {r.stdout.strip() or r.stderr.strip()[-300:]}", flush=True)

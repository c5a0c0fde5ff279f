#!/usr/bin/env python3
"""What this HIP runtime accepts of hipMemAddressReserve / hipMemMap (the measurements csrc/vmem.hip is designed around).
Talks to libamdhip64 directly (ctypes).  Every experiment runs in a FRESH process: the runtime keeps state across
reservations.  A sequence is piece sizes in MiB mapped back to back inside one reservation; '/' unmaps everything, frees the
reservation and starts a new one (which the runtime places at the same address).  After each reservation's pieces the
mapped range is filled and read back.
    python tools/vmem_probe.py                      all experiments
    python tools/vmem_probe.py one <reserve MiB> <piece> <piece> ... """
import ctypes
import os
import subprocess
import sys

MiB = 1 << 20


class Loc(ctypes.Structure):
    _fields_ = [("type", ctypes.c_int), ("id", ctypes.c_int)]


class Flags(ctypes.Structure):
    _fields_ = [("compressionType", ctypes.c_ubyte), ("gpuDirectRDMACapable", ctypes.c_ubyte), ("usage", ctypes.c_ushort)]


class Prop(ctypes.Structure):
    _fields_ = [("type", ctypes.c_int), ("requestedHandleType", ctypes.c_int), ("location", Loc),
                ("win32HandleMetaData", ctypes.c_void_p), ("allocFlags", Flags)]


class Access(ctypes.Structure):
    _fields_ = [("location", Loc), ("flags", ctypes.c_int)]


def one(reserve_mib, pieces):
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemAddressReserve.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_ulonglong]
    hip.hipMemCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.POINTER(Prop), ctypes.c_ulonglong]
    hip.hipMemMap.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_ulonglong]
    hip.hipMemSetAccess.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(Access), ctypes.c_size_t]
    hip.hipMemUnmap.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    hip.hipMemRelease.argtypes = [ctypes.c_void_p]
    hip.hipMemAddressFree.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    assert hip.hipSetDevice(0) == 0
    prop = Prop(type=1, requestedHandleType=0, location=Loc(1, 0))          # pinned, device 0
    acc = Access(Loc(1, 0), 3)                                              # read + write
    reserve = int(reserve_mib * MiB)
    out, base, mapped, fill = [], None, [], 1
    for p in list(pieces) + ["/"]:
        if p == "/":
            if base is not None:
                at = sum(sz for _, sz in mapped)
                if at:
                    fill += 1
                    hip.hipMemset(base, fill, at)
                    host = (ctypes.c_ubyte * 4096)()
                    good = True
                    for off in range(0, at, max(MiB, at // 64)):
                        hip.hipMemcpy(host, ctypes.c_void_p(base.value + off), 4096, 2)
                        good &= all(b == fill for b in host[::512])
                    out.append("rw-ok" if good else "RW-BAD")
                off = 0
                for h, sz in mapped:
                    hip.hipMemUnmap(ctypes.c_void_p(base.value + off), sz)
                    hip.hipMemRelease(h)
                    off += sz
                out.append(f"free={hip.hipMemAddressFree(base, reserve)}")
            base, mapped = None, []
            continue
        if base is None:
            base = ctypes.c_void_p()
            rc = hip.hipMemAddressReserve(ctypes.byref(base), reserve, 16 * MiB, None, 0)
            out.append(f"[reserve rc={rc} base=0x{base.value or 0:x} %16MiB={(base.value or 0) % (16 * MiB) // MiB}]")
        sz = int(float(p) * MiB)
        at = sum(s for _, s in mapped)
        h = ctypes.c_void_p()
        rc = hip.hipMemCreate(ctypes.byref(h), sz, ctypes.byref(prop), 0)
        if rc == 0:
            rc = hip.hipMemMap(ctypes.c_void_p(base.value + at), sz, 0, h, 0)
            if rc == 0:
                rc = hip.hipMemSetAccess(ctypes.c_void_p(base.value + at), sz, ctypes.byref(acc), 1)
            if rc != 0:
                hip.hipMemRelease(h)
        out.append(f"{p}:{'ok' if rc == 0 else 'FAIL(%d)' % rc}")
        if rc == 0:
            mapped.append((h, sz))
    print(" ".join(out))


if len(sys.argv) > 1 and sys.argv[1] == "one":
    one(float(sys.argv[2]), sys.argv[3:])
    sys.exit(0)

EXPERIMENTS = {
    "uniform 2 MiB x 8": (256, ["2"] * 8),
    "uniform 16 MiB x 8": (1024, ["16"] * 8),
    "uniform 3 MiB x 8 (not a power of two)": (256, ["3"] * 8),
    "2 then 30": (256, ["2", "30"]),
    "2 then 8 8 8 6": (256, ["2", "8", "8", "8", "6"]),
    "2 2 then 4": (256, ["2", "2", "4"]),
    "first odd then uniform: 5 then 2 2 2": (256, ["5", "2", "2", "2"]),
    "uniform 2, free, uniform 2": (256, ["2"] * 6 + ["/"] + ["2"] * 6),
    "uniform 2, free, uniform 8": (256, ["2"] * 6 + ["/"] + ["8"] * 6),
    "uniform 8, free, uniform 2": (256, ["8"] * 6 + ["/"] + ["2"] * 12),
    "uniform 16, free, uniform 16, free, uniform 16": (512, ["16"] * 4 + ["/"] + ["16"] * 8 + ["/"] + ["16"] * 3),
}
for name, (reserve, pieces) in EXPERIMENTS.items():
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "one", str(reserve)] + pieces, capture_output=True, text=True)
    print(f"{name}: {r.stdout.strip() or r.stderr.strip()[-300:]}", flush=True)

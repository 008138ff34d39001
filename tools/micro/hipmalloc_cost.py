#!/usr/bin/env python3
"""What a hipMalloc / hipFree of the index build's size costs: ONE block of n GB against n blocks of 4.5 GB, first time in the process and again.
python3 tools/micro/hipmalloc_cost.py"""
import ctypes
import time

hip = ctypes.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipFree.argtypes = [ctypes.c_void_p]


def alloc(sizes):
    t0 = time.perf_counter()
    ps = []
    for n in sizes:
        p = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(p), n) == 0
        ps.append(p)
    t1 = time.perf_counter()
    for p in ps:
        assert hip.hipFree(p) == 0
    t2 = time.perf_counter()
    return (t1 - t0) * 1e3, (t2 - t1) * 1e3


hip.hipSetDevice(0)
G = 1 << 30
for name, sizes in (("4 x 4.5 GB", [int(4.5 * G)] * 4), ("1 x 18 GB", [18 * G]), ("4 x 4.5 GB", [int(4.5 * G)] * 4), ("1 x 18 GB", [18 * G]),
                    ("1 x 27 GB", [27 * G]), ("6 x 4.5 GB", [int(4.5 * G)] * 6), ("1 x 27 GB", [27 * G])):
    a, f = alloc(sizes)
    print(f"{name}: malloc {a:.2f} ms, free {f:.2f} ms", flush=True)

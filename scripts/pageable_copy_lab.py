#!/usr/bin/env python
"""How fast does a PAGEABLE host buffer reach the device, by the call that carries it? (ss_process takes the scheduler's buffers as they
are: include/specscan.h.) 128 MiB (16 frames of 2^20 CF32 samples) and 16 MiB (128 frames of 65536 int8 samples), each through
  hipMemcpy (synchronous)            hipMemcpyAsync on the null stream + sync        hipMemcpyAsync on a non-blocking stream + sync
  hipMemcpy2DAsync (pitch = width)   hipHostRegister once + hipMemcpyAsync           torch's .copy_() for comparison
Run on the GPU box:  python scripts/pageable_copy_lab.py"""
import ctypes as C
import time

import numpy as np
import torch

hip = C.CDLL("libamdhip64.so")
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipMemcpy2DAsync.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p]
hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
H2D = 1
dev = torch.device("cuda", 0)
s_nb = C.c_void_p()
assert hip.hipStreamCreateWithFlags(C.byref(s_nb), 1) == 0
s_bl = C.c_void_p()
assert hip.hipStreamCreateWithFlags(C.byref(s_bl), 0) == 0
for mib in (128, 16):
    nbytes = mib << 20
    host = np.random.default_rng(0).integers(0, 255, nbytes, dtype=np.uint8)
    d = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    hp, dp = host.ctypes.data, d.data_ptr()

    def timed(fn, reps=6):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        return (time.perf_counter() - t0) / reps

    def f_sync():
        assert hip.hipMemcpy(dp, hp, nbytes, H2D) == 0

    def f_async(stream):
        def f():
            assert hip.hipMemcpyAsync(dp, hp, nbytes, H2D, stream) == 0
            assert hip.hipStreamSynchronize(stream) == 0
        return f

    def f_2d():
        rows = 16
        assert hip.hipMemcpy2DAsync(dp, nbytes // rows, hp, nbytes // rows, nbytes // rows, rows, H2D, s_nb) == 0
        assert hip.hipStreamSynchronize(s_nb) == 0

    ht = torch.from_numpy(host)

    def f_torch():
        d.copy_(ht)
        torch.cuda.synchronize()

    res = {"hipMemcpy": timed(f_sync), "hipMemcpyAsync(null stream)+sync": timed(f_async(None)), "hipMemcpyAsync(blocking stream)+sync": timed(f_async(s_bl)),
           "hipMemcpyAsync(non-blocking stream)+sync": timed(f_async(s_nb)), "hipMemcpy2DAsync(non-blocking stream)+sync": timed(f_2d), "torch copy_": timed(f_torch)}
    t0 = time.perf_counter()
    assert hip.hipHostRegister(hp, nbytes, 0) == 0
    t_reg = time.perf_counter() - t0
    res["hipHostRegister once"] = t_reg
    res["registered: hipMemcpyAsync(non-blocking)+sync"] = timed(f_async(s_nb))
    hip.hipHostUnregister(hp)
    print(f"{mib} MiB pageable -> device:")
    for k, v in res.items():
        print(f"   {k:48s} {v * 1e3:8.2f} ms  {nbytes / v / 1e9:6.1f} GB/s")


# ---- does destroying streams change what a stream created afterwards gets? (ss_process was ten times slower on a context created after
# another one had been destroyed: profiles/r06/s10_summary.txt) ----
import sys
nbytes = 128 << 20
host = np.random.default_rng(1).integers(0, 255, nbytes, dtype=np.uint8)
d = torch.empty(nbytes, dtype=torch.uint8, device=dev)
hp, dp = host.ctypes.data, d.data_ptr()
hip.hipStreamDestroy.argtypes = [C.c_void_p]


def copy_ms(stream, reps=4):
    hip.hipMemcpyAsync(dp, hp, nbytes, H2D, stream)
    hip.hipStreamSynchronize(stream)
    t0 = time.perf_counter()
    for _ in range(reps):
        assert hip.hipMemcpyAsync(dp, hp, nbytes, H2D, stream) == 0
        assert hip.hipStreamSynchronize(stream) == 0
    return (time.perf_counter() - t0) / reps * 1e3


def new_stream():
    st = C.c_void_p()
    assert hip.hipStreamCreateWithFlags(C.byref(st), 1) == 0
    return st


print("128 MiB pageable -> device on streams created at different times:")
a = new_stream()
print(f"   stream A (new): {copy_ms(a):.2f} ms")
others = [new_stream() for _ in range(5)]
for o in others:
    copy_ms(o, 1)
print(f"   stream A with five more streams alive: {copy_ms(a):.2f} ms; the fifth of them: {copy_ms(others[-1]):.2f} ms")
for o in others:
    hip.hipStreamDestroy(o)
print(f"   stream A after the five were destroyed: {copy_ms(a):.2f} ms")
b = new_stream()
print(f"   stream B, created after that: {copy_ms(b):.2f} ms")
hip.hipStreamDestroy(a)
c2 = new_stream()
print(f"   stream C, created after A was destroyed too: {copy_ms(c2):.2f} ms; stream B again: {copy_ms(b):.2f} ms; the null stream: {copy_ms(None):.2f} ms")

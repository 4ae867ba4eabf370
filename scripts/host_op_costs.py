#!/usr/bin/env python3
"""Host cost (us per call, min of 5 x 20 000) of the Python-side operations the example tick is made of."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch

from mppi_playground_amd import _capi
from mppi_playground_amd._pool import RowPool

dev = torch.device("cuda", 0)
lib = _capi.load()
a = torch.zeros(25, 2, device=dev)
s = torch.zeros(1, 26, 4, device=dev)
pool = RowPool((25, 2), dev, torch.float32)
raw = torch._C._cuda_getCurrentRawStream(0)


def bench(label, fn, n=20000):
    best = 1e9
    for _ in range(5):
        t = time.perf_counter()
        for _ in range(n):
            fn()
        best = min(best, (time.perf_counter() - t) / n)
    torch.cuda.synchronize()
    print(f"{label:48s} {best * 1e6:6.2f} us")


bench("torch.empty(25, 2, device=cuda)", lambda: torch.empty(25, 2, device=dev, dtype=torch.float32))
bench("torch.empty((), bool, device=cuda)", lambda: torch.empty((), device=dev, dtype=torch.bool))
bench("RowPool.take", lambda: pool.take(raw))
bench("torch.cuda.current_stream(dev).cuda_stream", lambda: torch.cuda.current_stream(dev).cuda_stream)
bench("torch._C._cuda_getCurrentRawStream(0)", lambda: torch._C._cuda_getCurrentRawStream(0))
bench("a[0, :]", lambda: a[0, :])
bench("a[0]", lambda: a[0])
bench("s[:, :, :2]", lambda: s[:, :, :2])
bench("s.squeeze(1)", lambda: s.squeeze(1))
bench("a.data_ptr()", lambda: a.data_ptr())
bench("a.is_contiguous()", lambda: a.is_contiguous())
bench("torch.is_tensor(a) and a.is_cuda", lambda: torch.is_tensor(a) and a.is_cuda)
bench("a.dtype == torch.float32", lambda: a.dtype == torch.float32)
bench("a.shape == (25, 2)", lambda: a.shape == (25, 2))
bench("lib.mppi_abi_version()", lambda: lib.mppi_abi_version())
bench("lib.mppi_device_count()", lambda: lib.mppi_device_count())
